"""The oracle restates reference FEM; the reference has no tests or golden vectors and cannot be built here
(PARITY UNPINNED, see oracle/fem_oracle.h).  These tests check the restatement against INDEPENDENT models written
from the algorithm's definition rather than from the reference's code:

  * banded Myers  (src/align.c:102-147)   vs  a cell-by-cell banded DP
  * 16-bit x8 form (src/align.c:149-277)  vs  the 32-bit scalar form  (SURVEY.md Appendix B probe)
  * seeding/filter (src/filter.c:146-223) vs  the closed form of SURVEY.md Appendix A.2
  * traceback      (src/align.c:279-544)  vs  re-scoring the CIGAR/MD against the two strings
  * index          (src/index.c:57-98)    vs  a dictionary of k-mer positions + the file size formula
"""
import os

import numpy as np
import pytest

from oracle import fem_oracle as fo
from tests import util

CODE = {65: 0, 97: 0, 67: 1, 99: 1, 71: 2, 103: 2, 84: 3, 116: 3}


def code(c):
    return CODE.get(c, 4)


# ---------------------------------------------------------------- Myers vs banded DP
def banded_dp(e, pattern, text):
    """Edit distance of `text` against pattern[i+j], j in [0,2e] per column i; free start, free end, strict band."""
    INF = 10 ** 6
    L = len(text)
    prev = [0] * (2 * e + 1)
    score_path = 0
    for i in range(L):
        cur = [INF] * (2 * e + 1)
        for j in range(2 * e + 1):
            d = prev[j] + (0 if code(text[i]) == code(pattern[i + j]) else 1)
            h = prev[j + 1] + 1 if j + 1 <= 2 * e else INF
            v = cur[j - 1] + 1 if j >= 1 else INF
            cur[j] = min(d, h, v)
        prev = cur
        score_path = cur[0]
        if score_path > 3 * e:
            return e + 1, None
    best, end = prev[0], L - 1
    for j in range(1, 2 * e + 1):
        if prev[j] < best:
            best, end = prev[j], L - 1 + j
    return best, end


@pytest.mark.parametrize("e", [0, 1, 2, 3, 5, 7])
def test_myers32_equals_banded_dp(e):
    rng = np.random.default_rng(100 + e)
    n_checked = 0
    for trial in range(250):
        L = int(rng.integers(30, 80))
        ref = util.rand_seq(rng, L + 4 * e + 8)
        shift = int(rng.integers(0, 2 * e + 1))
        read = util.mutate(rng, ref[shift:shift + L + e], int(rng.integers(0, e + 2)))[:L]
        if len(read) < L:
            continue
        if trial % 7 == 0:  # low-complexity + N
            read = bytearray(read)
            read[int(rng.integers(0, L))] = 78
            read = bytes(read)
        ed, end = fo.banded_ed32(e, ref, read)
        m_ed, m_end = banded_dp(e, ref, read)
        if m_end is None:
            assert ed == e + 1
        else:
            # the bit-vector form clamps nothing: scores agree exactly, as does the first strict minimum
            assert (ed, end) == (m_ed, m_end), (e, trial, ref, read)
        n_checked += 1
    assert n_checked > 200


@pytest.mark.parametrize("e", [1, 3, 7])
def test_myers16x8_agrees_with_myers32_on_accepts(e):
    rng = np.random.default_rng(7 + e)
    n_acc = 0
    for trial in range(120):
        L = int(rng.integers(40, 160))
        pats, reads_src = [], None
        base = util.rand_seq(rng, L + 4 * e + 8)
        read = util.mutate(rng, base[e:e + L + e], int(rng.integers(0, e + 1)))[:L]
        for lane in range(8):
            if rng.random() < 0.6:
                p = util.mutate(rng, base, int(rng.integers(0, 3)))
                p = (p + util.rand_seq(rng, 16))[:L + 4 * e + 8]
            else:
                p = util.rand_seq(rng, L + 4 * e + 8)
            pats.append(p)
        ed16, end16 = fo.banded_ed16x8(e, pats, read)
        for lane in range(8):
            ed32, end32 = fo.banded_ed32(e, pats[lane], read)
            assert (ed16[lane] <= e) == (ed32 <= e)
            if ed32 <= e:
                assert (int(ed16[lane]), int(end16[lane])) == (ed32, end32)
                n_acc += 1
    assert n_acc > 100


# ---------------------------------------------------------------- seeding vs the closed form
def closed_form_candidates(ref, idx, seq, e, a, k=12, s=3):
    L = len(seq)
    R = e + 1 + a
    lg = -(-k // s)
    S = L - k + 1
    if S <= 0 or R > S // s:
        return [], 0
    if (S - (s - 1)) // s - R * lg + 2 < 2:
        return [], 0
    mask = (1 << (2 * k)) - 1
    hashes = []
    for i in range(S):
        h = 0
        for c in seq[i:i + k]:
            h = ((h << 2) | (code(c) if code(c) < 4 else 0)) & mask
        hashes.append(h)
    if sum(1 for c in seq[k:] if code(c) == 4) > e:
        return [], 0
    lookup, occ = idx.lookup, idx.occ
    cand, pre = [], 0
    for si in range(s):
        G = (S - si) // s
        freq = [int(lookup[hashes[si + j * s] + 1]) - int(lookup[hashes[si + j * s]]) for j in range(G)]
        C = G - R * lg + 2
        INF = idx.n_occ & 0xFFFFFFFF
        M = [[0] * C for _ in range(R + 1)]
        D = [[3] * C for _ in range(R + 1)]
        for r in range(1, R + 1):
            M[r][0] = INF
            for c in range(1, C):
                w = (M[r - 1][c] + freq[c + (r - 1) * lg - 1]) & 0xFFFFFFFF
                if w < M[r][c - 1]:
                    M[r][c], D[r][c] = w, 2
                else:
                    M[r][c], D[r][c] = M[r][c - 1], 1
        pre += M[R][C - 1]
        picked = []
        r, c = R, C - 1
        while D[r][c] != 3:
            if D[r][c] == 2:
                picked.append(c + (r - 1) * lg - 1)
                r -= 1
            else:
                c -= 1
        picked += [None] * (R - len(picked))
        order = sorted(range(R), key=lambda t: (freq[picked[t]] if picked[t] is not None else 0, t))  # stable
        shifted = []
        for t in order:
            if picked[t] is None:
                shifted.append([])
                continue
            j = picked[t]
            start = si + j * s
            lo = int(lookup[hashes[start]])
            lst = [int(o) - start for o in occ[lo:lo + freq[j]] if (int(o) & 0xFFFFFFFF) >= start]
            shifted.append(lst)
        U = sorted(x for lst in shifted[:-1] for x in lst)
        T = [q for q in shifted[-1] if U and q <= U[-1]]
        X = sorted(U + T)
        F = [X[i] for i in range(len(X)) if i + a < len(X) and X[i + a] <= X[i] + e]
        merged = sorted(cand + F)
        cand = []
        for x in merged:
            if not cand or x > cand[-1] + e:
                cand.append(x)
    out = []
    for x in cand:
        sq, pos = x >> 32, x & 0xFFFFFFFF
        if pos >= e and pos + L + e < int(ref.len[sq]):
            out.append(x - e)
    return out, pre


@pytest.mark.parametrize("e,a,L", [(3, 1, 100), (2, 1, 80), (0, 0, 60), (7, 1, 150), (3, 2, 100), (3, 0, 100), (1, 1, 60)])
def test_seeding_equals_closed_form(e, a, L):
    rng = np.random.default_rng(1000 * e + 10 * a + L)
    seqs = util.repeat_rich_reference(rng, n_seq=2, unit_len=max(L + 60, 200), n_units=4, copies=25, spacer=120)
    ref = fo.Reference(seqs)
    idx = fo.OracleIndex(ref)
    reads = util.make_reads(rng, seqs, 60, L, e, n_rate=0.004)
    n_nonempty = 0
    for r in reads:
        for strand in (r, fo.revcomp(r)):
            got, pre = fo.seed_candidates(ref, idx, strand, e=e, a=a)
            want, wpre = closed_form_candidates(ref, idx, strand, e, a)
            assert pre == wpre
            assert list(map(int, got)) == want
            n_nonempty += bool(want)
    if e + a > 0:
        assert n_nonempty > 20
    else:  # R == 1: the only seed is also the truncated last seed (src/filter.c:85) -> never any candidate
        assert n_nonempty == 0


# ---------------------------------------------------------------- traceback re-scoring
def rescore(pattern, text, start, cigar, md):
    """Walk CIGAR over both strings; return (#edits, read bases consumed, MD rebuilt from first principles)."""
    import re
    ops = [(int(n), o) for n, o in re.findall(r"(\d+)([MID])", cigar)]
    rp, tp, edits = start, 0, 0
    md_parts, run = [], 0
    for n, o in ops:
        if o == "M":
            for _ in range(n):
                if pattern[rp] == text[tp]:
                    run += 1
                else:
                    edits += 1
                    if run:
                        md_parts.append(str(run))
                        run = 0
                    md_parts.append(chr(pattern[rp]))
                rp += 1
                tp += 1
        elif o == "I":
            tp += n
            edits += n
        else:
            if run:
                md_parts.append(str(run))
                run = 0
            md_parts.append("^" + pattern[rp:rp + n].decode())
            rp += n
            edits += n
    if run:
        md_parts.append(str(run))
    return edits, tp, "".join(md_parts), rp


@pytest.mark.parametrize("e", [1, 3, 7])
def test_traceback_cigar_has_exactly_ed_edits(e):
    rng = np.random.default_rng(55 + e)
    n = n_exact = 0
    for trial in range(400):
        L = int(rng.integers(50, 151))
        ref = util.rand_seq(rng, L + 4 * e + 8)
        read = util.mutate(rng, ref[e:e + L + e], int(rng.integers(0, e + 1)))[:L]
        if len(read) < L:
            continue
        ed, end = fo.banded_ed32(e, ref, read)
        if ed > e:
            continue
        start, cigar, md = fo.align(e, ref, read, ed, end)
        assert start >= 0, (ref, read, ed, end)
        edits, used, md_model, rp = rescore(ref, read, start, cigar, md)
        assert used == L
        assert md == md_model
        # read-end insertions are folded into the adjacent M run by the 'S' pseudo-op (src/align.c:358-365,
        # 466-469): the CIGAR then covers that many extra reference bases and may score fewer edits than NM.
        if rp - 1 == end:
            assert edits == ed, (cigar, md, ed)
            n_exact += 1
        else:
            assert rp - 1 > end and edits <= ed and "D" not in cigar.split("M")[-1]
        n += 1
    assert n > 250 and n_exact > 0.9 * n


# ---------------------------------------------------------------- sort: klib radix with ties
def test_mapping_sort_small_is_stable_and_large_is_sorted():
    rng = np.random.default_rng(9)
    for n in (0, 1, 5, 64, 65, 200, 1000):
        keys = rng.integers(0, 40, size=n).astype(np.uint64) << np.uint64(50)
        keys |= rng.integers(0, 4, size=n).astype(np.uint64)
        k2 = keys.copy()
        perm = np.zeros(n, np.uint32)
        fo.lib().fo_sort_mapping_keys(k2.ctypes.data, perm.ctypes.data, n)
        assert np.all(np.diff(k2.astype(np.int64)) >= 0)
        assert np.array_equal(keys[perm], k2)
        assert sorted(perm.tolist()) == list(range(n))
        if n <= 64:  # insertion sort path: stable
            assert np.array_equal(perm, np.argsort(keys, kind="stable").astype(np.uint32))


# ---------------------------------------------------------------- index
def test_index_layout_and_file_format(tmp_path):
    rng = np.random.default_rng(3)
    seqs = [util.rand_seq(rng, 5000), b"ACGTN" * 30 + util.rand_seq(rng, 777), util.rand_seq(rng, 13)]
    ref = fo.Reference(seqs)
    idx = fo.OracleIndex(ref)
    want = sum((len(s) - 12) // 3 + 1 for s in seqs if len(s) >= 12)
    assert idx.n_occ == want
    # every bucket ascending by location; contents match a dictionary of k-mers
    table = {}
    for si, s in enumerate(seqs):
        for pos in range(0, len(s) - 11, 3):
            h = 0
            for c in s[pos:pos + 12]:
                h = (h << 2) | (code(c) if code(c) < 4 else 0)
            table.setdefault(h, []).append((si << 32) | pos)
    for h, locs in table.items():
        lo, hi = int(idx.lookup[h]), int(idx.lookup[h + 1])
        assert idx.occ[lo:hi].tolist() == sorted(locs)
    assert int(idx.lookup[-1]) == want
    path = str(tmp_path / "t.idx")
    idx.save(path)
    # int32 k | int32 step | uint32 lookup[4^k+1] | size_t n | uint64 occ[n]
    assert os.path.getsize(path) == 8 + 4 * (4 ** 12 + 1) + 8 + 8 * want
    raw = np.fromfile(path, dtype=np.uint8)
    assert raw[:8].view(np.int32).tolist() == [12, 3]
    assert np.array_equal(raw[8:8 + 4 * (4 ** 12 + 1)].view(np.uint32), idx.lookup)
    assert int(raw[8 + 4 * (4 ** 12 + 1):][:8].view(np.uint64)[0]) == want
    assert np.array_equal(raw[8 + 4 * (4 ** 12 + 1) + 8:].view(np.uint64), idx.occ[:want])


def test_index_size_matches_survey_probe():
    # SURVEY.md §3.1 [probe]: 1 Mbp / step 3 -> n = 333 330, file 69 775 524 B (depends on length only)
    ref = fo.Reference([b"A" * 1_000_000])
    n = fo.lib().fo_index_count(__import__("ctypes").byref(ref.c), 12, 3)
    assert n == 333_330
    assert 8 + 4 * (4 ** 12 + 1) + 8 + 8 * n == 69_775_524


# ---------------------------------------------------------------- whole batch: threads agree, stats add up
def test_batch_driver_threads_and_stats():
    rng = np.random.default_rng(77)
    seqs = [util.rand_seq(rng, 60_000), util.rand_seq(rng, 40_000)]
    ref = fo.Reference(seqs)
    idx = fo.OracleIndex(ref)
    reads = fo.ReadBatch(util.make_reads(rng, seqs, 300, 100, 3))
    r1 = fo.map_reads(ref, idx, reads, e=3, threads=1)
    r4 = fo.map_reads(ref, idx, reads, e=3, threads=4)
    for f in ("stats", "cand_off", "cands", "pre", "v_ed", "v_end", "map_off", "m_dir", "m_ed", "m_cand", "m_end",
              "rec_off", "r_flag", "r_tid", "r_pos", "r_nm", "cig_off", "cig", "md_off", "md"):
        assert np.array_equal(getattr(r1, f), getattr(r4, f)), f
    assert r1.stats[0] == 300
    assert r1.stats[2] == r1.pre.sum()
    assert r1.stats[3] == len(r1.cands)
    assert r1.stats[4] == len(r1.m_cand) == len(r1.r_flag)
    assert r1.stats[1] == np.count_nonzero(np.diff(r1.map_off.astype(np.int64)))
    assert r1.stats[1] > 200  # most synthetic reads map
    assert not np.any(r1.r_flag & 0x8000)


def test_threaded_index_build_equals_the_sequential_restatement():
    # fo_index_build_mt only exists to make the checker usable on BASELINE-sized references (tests/test_gpu_full_scale.py);
    # it must give byte for byte what the restatement of construct_index (src/index.c:57-98) gives
    from tests import util
    rng = np.random.default_rng(31)
    seqs = [util.rand_seq(rng, 700_000), b"ACGT" * 5000, util.rand_seq(rng, 11), b"N" * 50 + util.rand_seq(rng, 90_000), b"A" * 3000]
    ref = fo.Reference(seqs)
    a = fo.OracleIndex(ref)
    for threads in (2, 5, 16):
        b = fo.OracleIndex(ref, threads=threads)
        assert a.n_occ == b.n_occ and np.array_equal(a.lookup, b.lookup) and np.array_equal(a.occ, b.occ)
