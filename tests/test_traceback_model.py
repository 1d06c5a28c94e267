"""An independent model of the reference's traceback (generate_alignment + generate_MD_tag, src/align.c:279-544), set
against the oracle's restatement of it (oracle/fem_oracle.c: fo_align).

The oracle keeps Myers' bit vectors per column as the reference does.  This model does not: it fills the banded edit
matrix cell by cell (cell (t, j) = read base t against pattern[t + j], j in [0, 2e]; free start, the cell above the band
unreachable, the cell below it one more than the band's last — what the bit vectors imply), reads the two bits the walk
asks for off the matrix — D0 = "the diagonal step costs nothing" (D[t][j] == D[t-1][j]), HP = "the horizontal step costs
one" (D[t][j] - D[t-1][j+1] == 1) — and then walks as src/align.c:340-479 does: match / mismatch / insertion / deletion
in that order of tests, the 'S' pseudo-run that folds read-end errors into the operation that follows, the stop at ed
errors, the remaining bases as M, operations emitted in reverse; the MD tag from the CIGAR (src/align.c:501-544).  It
was written from the reference source, not from the oracle's code, and shares nothing with it but the inputs.
Every emitted path is also re-scored: it must spend exactly `ed` edits unless read-end insertions were folded.
"""
import numpy as np
import pytest

from oracle import fem_oracle as fo
from tests import util

CODE = {65: 0, 97: 0, 67: 1, 99: 1, 71: 2, 103: 2, 84: 3, 116: 3}


def code(c):
    return CODE.get(c, 4)  # src/utils.h:72


def banded_matrix(e, pattern, text):
    """D[t][j], t in 0..L-1, j in 0..2e, and the column before the first (all zero: free start)."""
    INF = 10 ** 6
    W = 2 * e + 1
    prev = [0] * W
    cols = []
    for t in range(len(text)):
        cur = [INF] * W
        for j in range(W):
            diag = prev[j] + (0 if code(text[t]) == code(pattern[t + j]) else 1)
            left = prev[j + 1] + 1 if j + 1 < W else INF  # (t-1, q) with q below the previous column's band
            up = cur[j - 1] + 1 if j >= 1 else INF
            cur[j] = min(diag, left, up)
        cols.append(cur)
        prev = cur
    return cols


def bits(e, cols, t, j):
    """(D0, HP) of cell (t, j) as the walk reads them."""
    W = 2 * e + 1
    prev = cols[t - 1] if t > 0 else [0] * W
    d0 = cols[t][j] == prev[j]
    left = prev[j + 1] if j + 1 < W else prev[j] + 1  # below the band the vertical step costs one (VP's high bits are set)
    hp = cols[t][j] - left == 1
    return d0, hp


class Asserted(Exception):
    pass


def model_align(e, pattern, text, ed, end):
    """-> (start, [(op, len)] left to right, md string); Asserted where the reference would trip an assert."""
    L = len(text)
    start = end - L + 1
    if start < 0:
        raise Asserted("start")
    if all(text[i] == pattern[start + i] for i in range(L)):  # raw characters (src/align.c:289-300)
        cigar = [("M", L)]
        return start, cigar, model_md(pattern, text, start, cigar)
    cols = banded_matrix(e, pattern, text)
    bit, t, p = end - L + 1, L - 1, end
    nerr = 0
    d0, hp = bits(e, cols, t, bit)
    if d0 and pattern[p] == text[t]:
        t, p, pre, n = t - 1, p - 1, "M", 1
    elif not d0:
        if pattern[p] == text[t]:
            raise Asserted("mismatch on equal characters")
        t, p, nerr, pre, n = t - 1, p - 1, nerr + 1, "S", 1
    elif d0 and hp:
        t, bit, nerr, pre, n, start = t - 1, bit + 1, nerr + 1, "S", 1, start + 1
    else:
        raise Asserted("deletion first")
    ops = []
    while t >= 0 and nerr != ed:
        if bit < 0 or bit > 2 * e:
            raise Asserted("off the band")
        d0, hp = bits(e, cols, t, bit)
        if d0 and pattern[p] == text[t]:
            t, p = t - 1, p - 1
            if pre != "M":
                ops.append((pre, n))
                pre, n = "M", 1
            else:
                n += 1
        elif not d0:
            if pattern[p] == text[t]:
                raise Asserted("mismatch on equal characters")
            t, p, nerr = t - 1, p - 1, nerr + 1
            if pre == "S":
                n += 1
            elif pre != "M":
                ops.append((pre, n))
                pre, n = "M", 1
            else:
                n += 1
        elif d0 and hp:
            t, bit, nerr, start = t - 1, bit + 1, nerr + 1, start + 1
            if pre == "S":
                n += 1
            elif pre != "I":
                ops.append((pre, n))
                pre, n = "I", 1
            else:
                n += 1
        else:
            bit, p, nerr, start = bit - 1, p - 1, nerr + 1, start - 1
            if pre != "D":
                ops.append((pre, n))
                pre, n = "D", 1
            else:
                n += 1
    if t >= 0:
        if pre != "M":
            ops.append((pre, n))
            ops.append(("M", t + 1))
        else:
            ops.append(("M", n + t + 1))
    else:
        ops.append((pre, n))
    if ops[0][0] == "S":
        if len(ops) < 2:
            raise Asserted("only the pseudo-run")
        ops[1] = (ops[1][0], ops[1][1] + ops[0][1])
        ops = ops[1:]
    if any(op == "S" for op, _ in ops):
        raise Asserted("S inside")
    cigar = ops[::-1]
    if start < 0:
        raise Asserted("start")
    return start, cigar, model_md(pattern, text, start, cigar)


def model_md(pattern, text, start, cigar):
    out, run, rp, tp = [], 0, start, 0
    for op, n in cigar:
        if op == "M":
            for _ in range(n):
                if pattern[rp] == text[tp]:
                    run += 1
                else:
                    if run:
                        out.append(str(run))
                        run = 0
                    out.append(chr(pattern[rp]))
                rp, tp = rp + 1, tp + 1
        elif op == "I":
            tp += n
        else:
            if run:
                out.append(str(run))
                run = 0
            out.append("^" + pattern[rp:rp + n].decode("latin-1"))
            rp += n
    if run:
        out.append(str(run))
    return "".join(out)


def path_cost(pattern, text, start, cigar):
    """Edits the CIGAR spends (on base codes) and the reference position behind its last base."""
    rp, tp, cost = start, 0, 0
    for op, n in cigar:
        if op == "M":
            cost += sum(code(pattern[rp + i]) != code(text[tp + i]) for i in range(n))
            rp, tp = rp + n, tp + n
        elif op == "I":
            cost, tp = cost + n, tp + n
        else:
            cost, rp = cost + n, rp + n
    return cost, rp, tp


def cigar_str(cigar):
    return "".join("%d%s" % (n, op) for op, n in cigar)


@pytest.mark.parametrize("e", [1, 2, 3, 5, 7])
def test_model_walk_equals_the_oracle_traceback(e):
    rng = np.random.default_rng(900 + e)
    n = n_indel = n_folded = n_diagonal = 0
    for trial in range(700):
        L = int(rng.integers(30, 161))
        ref = util.rand_seq(rng, L + 4 * e + 8)
        shift = int(rng.integers(0, 2 * e + 1))
        read = util.mutate(rng, ref[shift:shift + L + e], int(rng.integers(0, e + 1)))[:L]
        if len(read) < L:
            continue
        if trial % 5 == 0:  # edits at the read's ends: the 'S' pseudo-run and the band's edges
            r = bytearray(read)
            for at in ((L - 1, L - 2) if trial % 10 else (0, 1)):
                r[at] = util.ACGT[(util.ACGT.tolist().index(r[at]) + 1) % 4] if r[at] in b"ACGT" else 65
            read = bytes(r)
        if trial % 9 == 0:  # low complexity
            k = int(rng.integers(0, L - 12))
            read = read[:k] + read[k:k + 3] * 4 + read[k + 12:]
            ref = ref[:shift + k] + read[k:k + 12] + ref[shift + k + 12:]
        if trial % 13 == 0:
            r = bytearray(read)
            r[int(rng.integers(0, L))] = 78
            read = bytes(r)
        ed, end = fo.banded_ed32(e, ref, read)
        if ed > e:
            continue
        o_start, o_cigar, o_md = fo.align(e, ref, read, ed, end)
        try:
            m_start, m_cig, m_md = model_align(e, ref, read, ed, end)
        except Asserted:
            assert o_start < 0, (ref, read, ed, end, o_start, o_cigar)
            continue
        assert o_start >= 0 and (o_start, o_cigar, o_md) == (m_start, cigar_str(m_cig), m_md), (e, trial, ref, read, ed, end)
        cost, rp, tp = path_cost(ref, read, m_start, m_cig)
        assert tp == L
        if rp - 1 == end:
            assert cost == ed  # a minimum-edit path: the matrix says no path to (L-1, end) is cheaper
        else:  # read-end insertions folded into the adjacent M run (src/align.c:358-365,466-469)
            assert rp - 1 > end and cost <= ed
            n_folded += 1
        n += 1
        n_indel += any(op != "M" for op, _ in m_cig)
        # what trace_ident_kernel relies on (fem_amd/csrc/fem_tail.hip): ed mismatching columns on the end position's
        # diagonal, characters standing for their codes -> the walk stays on that diagonal and emits `L M`
        d0 = end - L + 1
        canonical = all(c in b"ACGTN" for c in read) and all(c in b"ACGTN" for c in ref[d0:d0 + L]) if d0 >= 0 else False
        if canonical and L > ed and sum(code(ref[d0 + i]) != code(read[i]) for i in range(L)) == ed:
            assert m_cig == [("M", L)] and m_start == d0, (e, trial, ref, read, ed, end, m_cig)
            n_diagonal += 1
    assert n > 500 and n_indel > 60 and n_diagonal > 150, (n, n_indel, n_folded, n_diagonal)


def test_model_walk_equals_the_oracle_on_the_repeat_fixture_records():
    # every record of the committed repeat-rich fixture (tests/golden/repeat_rich.npz: 14 045 records, 387 of its reads
    # with indels, N runs in the reference), rebuilt from the accepted candidates by the model: the same multiset of
    # (strand, position, NM, CIGAR, MD) per read as the oracle's records
    from tests.golden.make_golden import CASES, inputs
    case = CASES["repeat_rich"]
    text, off, lens, bases, offs = inputs(case)
    seqs = [text[int(o):int(o) + int(l)].tobytes() for o, l in zip(off, lens)]
    ref = fo.Reference(seqs)
    idx = fo.OracleIndex(ref)
    res = fo.map_reads(ref, idx, fo.ReadBatch.from_arrays(bases, offs), e=case["e"], a=case["a"])
    e, n_checked = case["e"], 0
    for r in range(0, case["n_reads"], 3):  # a third of the reads: ~4 700 records
        read = bases[int(offs[r]):int(offs[r + 1])].tobytes()
        want = sorted((int(res.r_flag[j]) & 16, int(res.r_tid[j]), int(res.r_pos[j]), int(res.r_nm[j]), res.cigar_str(j), res.md_str(j))
                      for j in range(int(res.rec_off[r]), int(res.rec_off[r + 1])) if not int(res.r_flag[j]) & 0x8000)
        got = []
        for m in range(int(res.map_off[r]), int(res.map_off[r + 1])):
            d, ed, cand, end = int(res.m_dir[m]), int(res.m_ed[m]), int(res.m_cand[m]), int(res.m_end[m])
            sq, pos = cand >> 32, cand & 0xFFFFFFFF
            pattern = seqs[sq][pos:pos + len(read) + 2 * e]
            txt = fo.revcomp(read) if d else read
            try:
                start, cig, md = model_align(e, pattern, txt, ed, end)
            except Asserted:
                continue
            got.append((16 if d else 0, sq, pos + start, ed, cigar_str(cig), md))
        assert sorted(got) == want, r
        n_checked += len(want)
    assert n_checked > 3000
