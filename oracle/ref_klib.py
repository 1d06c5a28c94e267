"""ctypes binding of oracle/_ref/libfemref_klib.so — the reference's own kseq.h / ksort.h behind oracle/ref_klib.c.
TEST INFRASTRUCTURE ONLY: loaded by tests/ and tests/golden/make_klib_golden.py, never by the product.
The library is built by `make -C oracle ref` where /root/reference exists and ships to the GPU box as built."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class _Records(C.Structure):
    _fields_ = [("n", C.c_uint64), ("last_rc", C.c_int32), ("seq_off", C.c_void_p), ("name_off", C.c_void_p),
                ("comment_off", C.c_void_p), ("has_qual", C.c_void_p), ("seq", C.c_void_p), ("qual", C.c_void_p),
                ("name", C.c_void_p), ("comment", C.c_void_p)]


def path():
    return os.path.join(_HERE, "_ref", "libfemref_klib.so")


def available():
    return os.path.exists(path())


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(path())
        L.ref_kseq_read_all.argtypes = [C.c_char_p, C.POINTER(_Records)]
        L.ref_records_free.argtypes = [C.POINTER(_Records)]
        L.ref_radix_sort.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        _LIB = L
    return _LIB


def _bytes(ptr, n):
    return C.string_at(ptr, n) if ptr and n else b""


def kseq_records(file_path):
    """([(name, comment, seq, qual or None), ...], last_rc) as kseq_read yields them (zero-length records included)."""
    r = _Records()
    if lib().ref_kseq_read_all(file_path.encode(), C.byref(r)) != 0:
        raise FileNotFoundError(file_path)
    try:
        n = int(r.n)
        so = np.frombuffer(C.string_at(r.seq_off, 8 * (n + 1)), np.uint64)
        no = np.frombuffer(C.string_at(r.name_off, 8 * (n + 1)), np.uint64)
        co = np.frombuffer(C.string_at(r.comment_off, 8 * (n + 1)), np.uint64)
        hq = np.frombuffer(C.string_at(r.has_qual, n), np.uint8) if n else np.zeros(0, np.uint8)
        seq, qual = _bytes(r.seq, int(so[n])), _bytes(r.qual, int(so[n]))
        name, comment = _bytes(r.name, int(no[n])), _bytes(r.comment, int(co[n]))
        out = []
        for i in range(n):
            a, b = int(so[i]), int(so[i + 1])
            out.append((name[int(no[i]):int(no[i + 1])], comment[int(co[i]):int(co[i + 1])], seq[a:b],
                        qual[a:b] if hq[i] else None))
        return out, int(r.last_rc)
    finally:
        lib().ref_records_free(C.byref(r))


def radix_sort(keys):
    """(sorted keys, perm): perm[i] = index the record at rank i had before (the reference's radix_sort_mapping)."""
    k = np.ascontiguousarray(keys, dtype=np.uint64).copy()
    tags = np.zeros(len(k), np.uint32)
    lib().ref_radix_sort(k.ctypes.data, tags.ctypes.data, len(k))
    return k, tags
