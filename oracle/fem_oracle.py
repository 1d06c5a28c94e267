"""ctypes binding of the CPU oracle (oracle/fem_oracle.c).

TEST INFRASTRUCTURE ONLY — imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never by the product package fem_amd.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

STAGE_SEED, STAGE_VERIFY, STAGE_ALIGN = 1, 2, 4


class Params(C.Structure):
    _fields_ = [("k", C.c_int32), ("step", C.c_int32), ("e", C.c_int32), ("a", C.c_int32)]


class Ref(C.Structure):
    _fields_ = [("text", C.c_void_p), ("off", C.c_void_p), ("len", C.c_void_p), ("n_seq", C.c_uint32)]


class Index(C.Structure):
    _fields_ = [("k", C.c_int32), ("step", C.c_int32), ("lookup", C.c_void_p), ("n_occ", C.c_uint64),
                ("occ", C.c_void_p)]


class Reads(C.Structure):
    _fields_ = [("bases", C.c_void_p), ("off", C.c_void_p), ("n", C.c_uint64)]


def build(force=False):
    so = os.path.join(_HERE, "libfemoracle.so")
    src = os.path.join(_HERE, "fem_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libfemoracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.fo_index_count.restype = C.c_uint64
        L.fo_index_count.argtypes = [C.POINTER(Ref), C.c_int, C.c_int]
        L.fo_index_build.argtypes = [C.POINTER(Ref), C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.fo_index_build_mt.argtypes = [C.POINTER(Ref), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.fo_index_save.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p]
        L.fo_index_load.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p,
                                    C.POINTER(C.c_uint64), C.c_void_p, C.c_uint64]
        L.fo_hash_seed.restype = C.c_uint32
        L.fo_hash_seed.argtypes = [C.c_uint64, C.c_int, C.c_char_p, C.c_uint64]
        L.fo_revcomp.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p]
        L.fo_seed_candidates.restype = C.c_uint32
        L.fo_seed_candidates.argtypes = [C.POINTER(Params), C.c_char_p, C.c_uint32, C.POINTER(Ref), C.POINTER(Index),
                                         C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        L.fo_banded_ed32.argtypes = [C.c_int, C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_int)]
        L.fo_banded_ed16x8.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.c_char_p, C.c_int, C.c_void_p, C.c_void_p]
        L.fo_align.argtypes = [C.c_int, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                               C.POINTER(C.c_int), C.c_char_p, C.c_int]
        L.fo_sort_mapping_keys.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.fo_map.restype = C.c_void_p
        L.fo_map.argtypes = [C.POINTER(Params), C.POINTER(Ref), C.POINTER(Index), C.POINTER(Reads), C.c_int, C.c_int]
        L.fo_result_free.argtypes = [C.c_void_p]
        L.fo_result_stats.argtypes = [C.c_void_p, C.c_void_p]
        pp = C.POINTER(C.c_void_p)
        L.fo_result_candidates.restype = C.c_uint64
        L.fo_result_candidates.argtypes = [C.c_void_p, pp, pp, pp]
        L.fo_result_verify.argtypes = [C.c_void_p, pp, pp]
        L.fo_result_mappings.restype = C.c_uint64
        L.fo_result_mappings.argtypes = [C.c_void_p, pp, pp, pp, pp, pp]
        L.fo_result_records.restype = C.c_uint64
        L.fo_result_records.argtypes = [C.c_void_p] + [pp] * 9
        _LIB = L
    return _LIB


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _view(ptr, n, dtype):
    """Copy n items of dtype from a C pointer into a fresh numpy array."""
    n = int(n)
    if n == 0 or not ptr.value:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr.value)
    return np.frombuffer(buf, dtype=dtype, count=n).copy()


class Reference:
    """Concatenated reference text (raw FASTA characters) + per-sequence offsets/lengths."""

    def __init__(self, seqs, names=None):
        seqs = [s if isinstance(s, (bytes, bytearray)) else bytes(s) for s in seqs]
        self.names = names or ["chr%d" % (i + 1) for i in range(len(seqs))]
        self.len = np.array([len(s) for s in seqs], dtype=np.uint32)
        self.off = np.zeros(len(seqs), dtype=np.uint64)
        if len(seqs) > 1:
            self.off[1:] = np.cumsum(self.len[:-1], dtype=np.uint64)
        # 64 bytes of slack: the reference reads pattern[] from malloc'd text and never past
        # candidate+L+2e < seq end, but keep the concatenation safely padded anyway
        self.text = np.frombuffer(b"".join(seqs) + b"\0" * 64, dtype=np.uint8).copy()
        self.c = Ref(_ptr(self.text), _ptr(self.off), _ptr(self.len), len(seqs))

    def seq(self, i):
        o = int(self.off[i])
        return self.text[o:o + int(self.len[i])].tobytes()


class ReadBatch:
    def __init__(self, reads):
        reads = [r if isinstance(r, (bytes, bytearray)) else bytes(r) for r in reads]
        self.n = len(reads)
        self.off = np.zeros(self.n + 1, dtype=np.uint64)
        if self.n:
            self.off[1:] = np.cumsum([len(r) for r in reads], dtype=np.uint64)
        self.bases = np.frombuffer(b"".join(reads) + b"\0" * 8, dtype=np.uint8).copy()
        self.c = Reads(_ptr(self.bases), _ptr(self.off), self.n)

    @classmethod
    def from_arrays(cls, bases, off):
        self = cls.__new__(cls)
        self.n = len(off) - 1
        self.off = np.ascontiguousarray(off, dtype=np.uint64)
        self.bases = np.ascontiguousarray(bases, dtype=np.uint8)
        self.c = Reads(_ptr(self.bases), _ptr(self.off), self.n)
        return self

    def read(self, i):
        return self.bases[int(self.off[i]):int(self.off[i + 1])].tobytes()


class OracleIndex:
    def __init__(self, ref, k=12, step=3, threads=1):
        L = lib()
        self.k, self.step = k, step
        n = L.fo_index_count(C.byref(ref.c), k, step)
        self.lookup = np.zeros((1 << (2 * k)) + 1, dtype=np.uint32)
        self.occ = np.zeros(max(int(n), 1), dtype=np.uint64)
        self.n_occ = int(n)
        if threads > 1:  # same arrays, several threads (BASELINE-sized references)
            rc = L.fo_index_build_mt(C.byref(ref.c), k, step, _ptr(self.lookup), _ptr(self.occ), threads)
        else:
            rc = L.fo_index_build(C.byref(ref.c), k, step, _ptr(self.lookup), _ptr(self.occ))
        assert rc == 0
        self.c = Index(k, step, _ptr(self.lookup), self.n_occ, _ptr(self.occ))

    @classmethod
    def from_arrays(cls, k, step, lookup, occ, n_occ=None):
        self = cls.__new__(cls)
        self.k, self.step = k, step
        self.lookup = np.ascontiguousarray(lookup, dtype=np.uint32)
        self.occ = np.ascontiguousarray(occ, dtype=np.uint64)
        self.n_occ = int(len(occ) if n_occ is None else n_occ)
        self.c = Index(k, step, _ptr(self.lookup), self.n_occ, _ptr(self.occ))
        return self

    def save(self, path):
        rc = lib().fo_index_save(path.encode(), self.k, self.step, _ptr(self.lookup), self.n_occ, _ptr(self.occ))
        assert rc == 0


class MapResult:
    """All outputs of fo_map copied into numpy arrays."""

    def __init__(self, h, n_reads):
        L = lib()
        st = np.zeros(5, dtype=np.uint64)
        L.fo_result_stats(h, _ptr(st))
        self.stats = st
        a, b, c, d, e, f, g, hh, i = (C.c_void_p() for _ in range(9))
        nc = L.fo_result_candidates(h, C.byref(a), C.byref(b), C.byref(c))
        self.cand_off = _view(a, 2 * n_reads + 1, np.uint64)
        self.cands = _view(b, nc, np.uint64)
        self.pre = _view(c, 2 * n_reads, np.uint32)
        L.fo_result_verify(h, C.byref(a), C.byref(b))
        self.v_ed = _view(a, nc, np.uint8)
        self.v_end = _view(b, nc, np.int16)
        if len(self.v_ed) != nc:  # verify stage not run
            self.v_ed = np.zeros(0, np.uint8)
            self.v_end = np.zeros(0, np.int16)
        nm = L.fo_result_mappings(h, C.byref(a), C.byref(b), C.byref(c), C.byref(d), C.byref(e))
        self.map_off = _view(a, n_reads + 1, np.uint64)
        self.m_dir = _view(b, nm, np.uint8)
        self.m_ed = _view(c, nm, np.uint8)
        self.m_cand = _view(d, nm, np.uint64)
        self.m_end = _view(e, nm, np.int16)
        nr = L.fo_result_records(h, C.byref(a), C.byref(b), C.byref(c), C.byref(d), C.byref(e), C.byref(f),
                                 C.byref(g), C.byref(hh), C.byref(i))
        self.rec_off = _view(a, n_reads + 1, np.uint64)
        self.r_flag = _view(b, nr, np.uint16)
        self.r_tid = _view(c, nr, np.uint32)
        self.r_pos = _view(d, nr, np.uint32)
        self.r_nm = _view(e, nr, np.uint8)
        self.cig_off = _view(f, nr + 1, np.uint64)
        self.cig = _view(g, int(self.cig_off[-1]) if nr else 0, np.uint32)
        self.md_off = _view(hh, nr + 1, np.uint64)
        self.md = _view(i, int(self.md_off[-1]) if nr else 0, np.uint8)

    def cigar_str(self, j):
        ops = self.cig[int(self.cig_off[j]):int(self.cig_off[j + 1])]
        return "".join("%d%s" % (int(o) >> 4, "MID"[int(o) & 0xF]) for o in ops)

    def md_str(self, j):
        return self.md[int(self.md_off[j]):int(self.md_off[j + 1])].tobytes().decode()


def map_reads(ref, index, reads, e=3, a=1, k=12, step=3, threads=1, stages=STAGE_SEED | STAGE_VERIFY | STAGE_ALIGN,
              keep_handle=False):
    p = Params(k, step, e, a)
    h = lib().fo_map(C.byref(p), C.byref(ref.c), C.byref(index.c), C.byref(reads.c), threads, stages)
    if keep_handle:
        return h
    try:
        return MapResult(C.c_void_p(h), reads.n)
    finally:
        lib().fo_result_free(C.c_void_p(h))


def free_result(h):
    lib().fo_result_free(C.c_void_p(h))


def banded_ed32(e, pattern, text):
    end = C.c_int(-len(text))
    ed = lib().fo_banded_ed32(e, pattern, text, len(text), C.byref(end))
    return ed, end.value


def banded_ed16x8(e, patterns, text):
    arr = (C.c_char_p * 8)(*patterns)
    ed = np.zeros(8, np.int16)
    end = np.full(8, len(text) - 1, np.int16)
    lib().fo_banded_ed16x8(e, arr, text, len(text), _ptr(ed), _ptr(end))
    return ed, end


def align(e, pattern, text, ed, end):
    cig = np.zeros(len(text) + 2, np.uint32)
    n = C.c_int(0)
    md = C.create_string_buffer(16 * len(text) + 64)
    start = lib().fo_align(e, pattern, text, len(text), ed, end, _ptr(cig), len(cig), C.byref(n), md, len(md))
    cigar = "".join("%d%s" % (int(o) >> 4, "MID"[int(o) & 0xF]) for o in cig[:n.value])
    return start, cigar, md.value.decode()


def revcomp(seq):
    out = C.create_string_buffer(len(seq))
    lib().fo_revcomp(seq, len(seq), out)
    return out.raw


def seed_candidates(ref, index, seq, e=3, a=1, k=12, step=3):
    p = Params(k, step, e, a)
    cap = 1 << 12
    while True:
        buf = np.zeros(cap, np.uint64)
        pre = C.c_uint32(0)
        n = lib().fo_seed_candidates(C.byref(p), seq, len(seq), C.byref(ref.c), C.byref(index.c), _ptr(buf), cap,
                                     C.byref(pre))
        if n != 0xFFFFFFFF:
            return buf[:n].copy(), pre.value
        cap *= 8
