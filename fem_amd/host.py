"""ctypes binding of libfemhost.so (fem_amd/csrc/fem_host.h): sequence/index files, mapping tail, synthetic data."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class SeqSet(C.Structure):
    _fields_ = [("n", C.c_uint64), ("bases", C.c_void_p), ("off", C.c_void_p), ("quals", C.c_void_p),
                ("names", C.c_void_p), ("name_off", C.c_void_p)]


class TailInput(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("cand_begin", C.c_void_p), ("cand_count", C.c_void_p), ("cand", C.c_void_p),
                ("ed", C.c_void_p), ("end", C.c_void_p)]


class TailRef(C.Structure):
    _fields_ = [("text", C.c_void_p), ("off", C.c_void_p), ("len", C.c_void_p), ("n_seq", C.c_uint32),
                ("names", C.c_void_p), ("name_off", C.c_void_p)]


class Records(C.Structure):
    _fields_ = [("n_records", C.c_uint64), ("rec_off", C.c_void_p), ("flag", C.c_void_p), ("tid", C.c_void_p),
                ("pos0", C.c_void_p), ("nm", C.c_void_p), ("cigar_off", C.c_void_p), ("cigar", C.c_void_p),
                ("md_off", C.c_void_p), ("md", C.c_void_p)]


class BatchShape(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("n_bases", C.c_uint64), ("n_name_bytes", C.c_uint64), ("max_len", C.c_uint32),
                ("has_qual", C.c_int32), ("min_len", C.c_uint32)]


class RecordView(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("n_records", C.c_uint64), ("rec_begin", C.c_void_p), ("flag", C.c_void_p),
                ("tid", C.c_void_p), ("pos0", C.c_void_p), ("nm", C.c_void_p), ("cigar_off", C.c_void_p),
                ("cigar", C.c_void_p), ("md_off", C.c_void_p), ("md", C.c_void_p)]


class TextPart(C.Structure):
    _fields_ = [("offset", C.c_uint64), ("length", C.c_uint64)]


def library_path():
    return os.path.join(_HERE, "csrc", "libfemhost.so")


def lib():
    global _LIB
    if _LIB is None:
        path = library_path()
        if not os.path.exists(path):
            raise RuntimeError("libfemhost.so is missing (%s): run `make -C fem_amd/csrc`" % path)
        L = C.CDLL(path)
        vp, u64, i32 = C.c_void_p, C.c_uint64, C.c_int32
        L.fem_seqfile_open.restype = vp
        L.fem_seqfile_open.argtypes = [C.c_char_p]
        L.fem_seqfile_close.argtypes = [vp]
        L.fem_seqfile_read.argtypes = [vp, u64, C.POINTER(SeqSet)]
        L.fem_seqset_free.argtypes = [C.POINTER(SeqSet)]
        L.fem_seqfile_read_bytes.argtypes = [vp, u64, C.c_int, C.POINTER(SeqSet)]
        L.fem_seqfile_plan.argtypes = [vp, u64, C.c_int, C.POINTER(vp), C.POINTER(BatchShape)]
        L.fem_seqfile_fill.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp, vp]
        L.fem_seqfile_fill_packed.argtypes = [vp, vp, C.c_int, C.c_uint32, vp, u64, C.POINTER(u64), vp, vp, vp]
        L.fem_seqfile_fill_packed_refs.argtypes = [vp, vp, C.c_int, C.c_uint32, vp, u64, C.POINTER(u64), vp]
        L.fem_batch_plan_free.argtypes = [vp]
        L.fem_records_sam.argtypes = [C.POINTER(TailRef), C.POINTER(SeqSet), C.POINTER(RecordView), C.c_int, C.POINTER(vp),
                                      C.POINTER(u64)]
        L.fem_records_sam_parts.argtypes = [C.POINTER(TailRef), C.POINTER(SeqSet), C.POINTER(RecordView), C.c_int,
                                            C.POINTER(vp), C.POINTER(u64), C.POINTER(TextPart), C.POINTER(u64)]
        L.fem_index_save.argtypes = [C.c_char_p, i32, i32, vp, u64, vp]
        L.fem_index_load.argtypes = [C.c_char_p, C.POINTER(i32), C.POINTER(i32), C.POINTER(vp), C.POINTER(u64),
                                     C.POINTER(vp)]
        L.fem_tail_records.argtypes = [i32, C.POINTER(TailRef), vp, vp, C.POINTER(TailInput), C.c_int,
                                       C.POINTER(Records)]
        L.fem_records_free.argtypes = [C.POINTER(Records)]
        L.fem_tail_sam.argtypes = [i32, C.POINTER(TailRef), C.POINTER(SeqSet), C.POINTER(TailInput), C.c_int,
                                   C.POINTER(vp), C.POINTER(u64)]
        L.fem_sam_header.argtypes = [C.POINTER(TailRef), C.POINTER(vp), C.POINTER(u64)]
        L.fem_sam_fill_quals.argtypes = [vp, u64, vp, u64, vp, vp, C.c_uint32, C.c_int]
        L.fem_synth_reference.argtypes = [u64, C.c_uint32, vp, vp, vp, C.c_int]
        L.fem_synth_reads.argtypes = [u64, vp, vp, vp, C.c_uint32, u64, u64, C.c_uint32, i32, vp, C.c_int]
        L.fem_synth_reads_ex.argtypes = [u64, vp, vp, vp, C.c_uint32, u64, u64, C.c_uint32, i32, vp, vp, C.c_int]
        L.fem_synth_reads_packed.argtypes = [u64, vp, vp, vp, C.c_uint32, u64, u64, C.c_uint32, i32, vp, C.c_int]
        L.fem_synth_write_fastq.argtypes = [C.c_char_p, vp, C.c_uint32, u64, u64]
        L.fem_synth_write_fasta.argtypes = [C.c_char_p, vp, vp, vp, C.c_uint32]
        L.free = C.CDLL(None).free
        L.free.argtypes = [vp]
        _LIB = L
    return _LIB


def _copy(ptr, n, dtype):
    n = int(n)
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=n).copy()


# ------------------------------------------------------------------------------------------ synthetic data
def synth_reference(seed, seq_lens, threads=8):
    """Concatenated iid A/C/G/T text for sequences of the given lengths -> (text uint8[total+64], off, len)."""
    lens = np.asarray(seq_lens, dtype=np.uint32)
    off = np.zeros(len(lens), dtype=np.uint64)
    if len(lens) > 1:
        off[1:] = np.cumsum(lens[:-1], dtype=np.uint64)
    total = int(lens.astype(np.uint64).sum())
    text = np.zeros(total + 64, dtype=np.uint8)
    lib().fem_synth_reference(seed, len(lens), off.ctypes.data, lens.ctypes.data, text.ctypes.data, threads)
    return text, off, lens


def synth_reads(seed, text, off, lens, n_reads, L, e, first_read=0, threads=8, out=None, out_offsets=None, n_err=None):
    """n_reads reads of length L drawn from the reference with 0..e edits -> (bases uint8[n*L(+8)], offsets uint64[n+1]).
    Uniform start, edit count uniform in 0..e, each edit 60 % substitution / 20 % insertion / 20 % deletion at a uniform
    interior offset of the read, 50 % reverse-complemented.  `out` / `out_offsets`: write into these
    arrays (e.g. the pinned staging views of Device.acquire_stage); `n_err`: uint8[n_reads] receiving the edit counts."""
    bases = np.zeros(n_reads * L + 8, dtype=np.uint8) if out is None else out
    assert bases.dtype == np.uint8 and len(bases) >= n_reads * L
    ne = 0
    if n_err is not None:
        assert n_err.dtype == np.uint8 and len(n_err) >= n_reads
        ne = n_err.ctypes.data
    lib().fem_synth_reads_ex(seed, text.ctypes.data, off.ctypes.data, lens.ctypes.data, len(lens), first_read, n_reads,
                             L, e, bases.ctypes.data, ne, threads)
    if out_offsets is None:
        offsets = np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(L)
    else:
        offsets = out_offsets
        offsets[:n_reads + 1] = np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(L)
    return bases, offsets


def synth_reads_packed(seed, text, off, lens, n_reads, L, e, out, first_read=0, threads=8):
    """The reads of synth_reads at two bits per base, written into `out` (uint8; e.g. the staging view of
    Device.acquire_stage): ceil(L / 4) bytes per read, zero-padded to a multiple of 8.  No exceptions (A C G T only)."""
    bpr = (L + 3) // 4
    assert out.dtype == np.uint8 and len(out) >= ((n_reads * bpr + 7) & ~7)
    lib().fem_synth_reads_packed(seed, text.ctypes.data, off.ctypes.data, lens.ctypes.data, len(lens), first_read, n_reads,
                                 L, e, out.ctypes.data, threads)
    return out


def write_fastq(path, bases, L, n_reads, first_index=0):
    rc = lib().fem_synth_write_fastq(path.encode(), bases.ctypes.data, L, n_reads, first_index)
    if rc != 0:
        raise OSError("fem_synth_write_fastq failed (%d)" % rc)


def write_fasta(path, text, off, lens):
    rc = lib().fem_synth_write_fasta(path.encode(), text.ctypes.data, off.ctypes.data, lens.ctypes.data, len(lens))
    if rc != 0:
        raise OSError("fem_synth_write_fasta failed (%d)" % rc)


# ------------------------------------------------------------------------------------------ files
class Sequences:
    """Host copy of a fem_seqset."""

    def __init__(self, s):
        n = int(s.n)
        self.n = n
        self.off = _copy(s.off, n + 1, np.uint64)
        self.name_off = _copy(s.name_off, n + 1, np.uint64)
        nb = int(self.off[-1]) if n else 0
        self.bases = _copy(s.bases, nb, np.uint8)
        self.quals = _copy(s.quals, nb, np.uint8) if s.quals else None
        self.names_raw = _copy(s.names, int(self.name_off[-1]) if n else 0, np.uint8)

    def seq(self, i):
        return self.bases[int(self.off[i]):int(self.off[i + 1])].tobytes()

    def name(self, i):
        return self.names_raw[int(self.name_off[i]):int(self.name_off[i + 1])].tobytes().decode()

    def qual(self, i):
        return self.quals[int(self.off[i]):int(self.off[i + 1])].tobytes()


def read_sequences(path, max_seqs=0):
    L = lib()
    f = L.fem_seqfile_open(path.encode())
    if not f:
        raise FileNotFoundError(path)
    s = SeqSet()
    rc = L.fem_seqfile_read(f, max_seqs, C.byref(s))
    L.fem_seqfile_close(f)
    try:
        if rc != 0:
            raise ValueError("malformed sequence file %s (rc=%d)" % (path, rc))
        return Sequences(s)
    finally:
        L.fem_seqset_free(C.byref(s))


def read_sequences_in_chunks(path, approx_bytes, threads=4):
    """All records of a file through fem_seqfile_read_bytes (the CLI's batch reader), one Sequences per batch."""
    L = lib()
    f = L.fem_seqfile_open(path.encode())
    if not f:
        raise FileNotFoundError(path)
    out = []
    try:
        while True:
            s = SeqSet()
            rc = L.fem_seqfile_read_bytes(f, approx_bytes, threads, C.byref(s))
            try:
                if rc != 0:
                    raise ValueError("malformed sequence file %s (rc=%d)" % (path, rc))
                if s.n == 0:
                    break
                out.append(Sequences(s))
            finally:
                L.fem_seqset_free(C.byref(s))
    finally:
        L.fem_seqfile_close(f)
    return out


class PlannedBatch:
    """One batch read in two phases (fem_seqfile_plan + fem_seqfile_fill) into caller-owned arrays."""

    def __init__(self, shape, bases, off, quals, names, name_off):
        self.n = int(shape.n_reads)
        self.max_len, self.min_len = int(shape.max_len), int(shape.min_len)
        self.bases, self.off, self.quals, self.names_raw, self.name_off = bases, off, quals, names, name_off

    def seq(self, i):
        return self.bases[int(self.off[i]):int(self.off[i + 1])].tobytes()

    def qual(self, i):
        return self.quals[int(self.off[i]):int(self.off[i + 1])].tobytes()

    def name(self, i):
        return self.names_raw[int(self.name_off[i]):int(self.name_off[i + 1])].tobytes().decode()


class PackedBatch:
    """One batch read by fem_seqfile_plan + fem_seqfile_fill_packed: `codes` holds the staging layout of
    fem_dev_commit_stage_packed (codes, exception positions, exception bytes)."""

    def __init__(self, shape, codes, n_exc, quals, names, name_off):
        self.n = int(shape.n_reads)
        self.read_len = int(shape.max_len)
        self.codes, self.n_exc, self.quals, self.names_raw, self.name_off = codes, int(n_exc), quals, names, name_off

    def unpack(self):
        """The batch's characters, rebuilt on the host as the device rebuilds them (tests)."""
        n, L = self.n, self.read_len
        bpr = (L + 3) // 4
        c = self.codes[:n * bpr].reshape(n, bpr)
        q = np.stack([(c >> (2 * j)) & 3 for j in range(4)], axis=2).reshape(n, bpr * 4)[:, :L]
        out = np.frombuffer(b"ACGT", np.uint8)[q].reshape(-1).copy()
        exc_off = (n * bpr + 7) & ~7
        pos = self.codes[exc_off:exc_off + 4 * self.n_exc].view(np.uint32)
        out[pos] = self.codes[exc_off + 4 * self.n_exc:exc_off + 5 * self.n_exc]
        return out


class _ReadRefs(C.Structure):
    _fields_ = [("n", C.c_uint64), ("read_len", C.c_uint32), ("name", C.c_void_p), ("name_len", C.c_void_p), ("seq", C.c_void_p),
                ("qual", C.c_void_p)]


def read_refs_batches(path, approx_bytes, threads=4, exc_cap=None):
    """The splice form of the command line (fem_seqfile_fill_packed_refs): per batch (rc, records) where rc is what the call
    returned — 0: the fields were read through the pointers it filled (into the file's mapping) and the bases also unpacked
    from the codes; 1 (too many characters outside ACGT) / 2 (the records do not sit in a mapping that stays): the plan was
    left alone and fem_seqfile_fill took the batch.  records = list of (name, seq, qual), seq twice where rc == 0."""
    L = lib()
    f = L.fem_seqfile_open(path.encode())
    if not f:
        raise FileNotFoundError(path)
    out = []
    try:
        while True:
            plan, shape = C.c_void_p(), BatchShape()
            rc = L.fem_seqfile_plan(f, approx_bytes, threads, C.byref(plan), C.byref(shape))
            if rc != 0:
                if plan:
                    L.fem_batch_plan_free(plan)
                raise ValueError("malformed sequence file %s (rc=%d)" % (path, rc))
            if shape.n_reads == 0:
                L.fem_batch_plan_free(plan)
                break
            n, nb = int(shape.n_reads), int(shape.n_bases)
            rc = 1
            if shape.min_len == shape.max_len:
                rl = int(shape.max_len)
                bpr = (rl + 3) // 4
                exc_off = (n * bpr + 7) & ~7
                cap = min((nb + 64 - min(exc_off, nb + 64)) // 5, nb // 16) if exc_cap is None else exc_cap
                codes = np.zeros(nb + 64, np.uint8)
                name, seq, qual = (np.zeros(n, np.uint64) for _ in range(3))
                name_len = np.zeros(n, np.uint32)
                refs = _ReadRefs(0, 0, name.ctypes.data, name_len.ctypes.data, seq.ctypes.data, qual.ctypes.data)
                n_exc = C.c_uint64()
                rc = L.fem_seqfile_fill_packed_refs(f, plan, threads, rl, codes.ctypes.data, cap, C.byref(n_exc), C.byref(refs))
                if rc == 0:
                    assert refs.n == n and refs.read_len == rl
                    unpacked = PackedBatch(shape, codes, n_exc.value, None, None, None).unpack().tobytes()
                    recs = [(C.string_at(int(name[i]), int(name_len[i])), C.string_at(int(seq[i]), rl), C.string_at(int(qual[i]), rl),
                             unpacked[i * rl:(i + 1) * rl]) for i in range(n)]
                    out.append((0, recs))
                    continue
                if rc not in (1, 2):
                    raise ValueError("fem_seqfile_fill_packed_refs failed (%d)" % rc)
            bases, off = np.zeros(nb + 64, np.uint8), np.zeros(n + 1, np.uint64)
            quals = np.zeros(nb + 1, np.uint8)
            names = np.zeros(int(shape.n_name_bytes) + 1, np.uint8)
            name_off = np.zeros(n + 1, np.uint64)
            if L.fem_seqfile_fill(f, plan, threads, bases.ctypes.data, off.ctypes.data, quals.ctypes.data, names.ctypes.data, name_off.ctypes.data) != 0:
                raise ValueError("fem_seqfile_fill failed")
            b = PlannedBatch(shape, bases, off, quals, names, name_off)
            as_bytes = lambda x: x.encode() if isinstance(x, str) else bytes(x)
            out.append((rc, [(as_bytes(b.name(i)), as_bytes(b.seq(i)), as_bytes(b.qual(i))) for i in range(n)]))
    finally:
        L.fem_seqfile_close(f)
    return out


def read_planned_batches(path, approx_bytes, threads=4, alloc=None, packed=False):
    """All records of a file through the two-phase reader the command line uses; `alloc(n_reads, n_bases)` may hand
    out the (bases uint8[n_bases + 64], off uint64[n_reads + 1]) arrays (e.g. pinned staging views).  packed=True:
    batches of equal-length reads come back as PackedBatch (fem_seqfile_fill_packed), others as PlannedBatch."""
    L = lib()
    f = L.fem_seqfile_open(path.encode())
    if not f:
        raise FileNotFoundError(path)
    out = []
    try:
        while True:
            plan, shape = C.c_void_p(), BatchShape()
            rc = L.fem_seqfile_plan(f, approx_bytes, threads, C.byref(plan), C.byref(shape))
            if rc != 0:
                if plan:
                    L.fem_batch_plan_free(plan)
                raise ValueError("malformed sequence file %s (rc=%d)" % (path, rc))
            if shape.n_reads == 0:
                L.fem_batch_plan_free(plan)
                break
            n, nb = int(shape.n_reads), int(shape.n_bases)
            if alloc:
                bases, off = alloc(n, nb)
            else:
                bases, off = np.zeros(nb + 64, np.uint8), np.zeros(n + 1, np.uint64)
            quals = np.zeros(nb + 1, np.uint8) if shape.has_qual else None
            names = np.zeros(int(shape.n_name_bytes) + 1, np.uint8)
            name_off = np.zeros(n + 1, np.uint64)
            if packed and shape.min_len == shape.max_len:
                bpr = (int(shape.max_len) + 3) // 4
                exc_off = (n * bpr + 7) & ~7
                exc_cap = min((nb + 64 - min(exc_off, nb + 64)) // 5, nb // 16)
                n_exc = C.c_uint64()
                rc = L.fem_seqfile_fill_packed(f, plan, threads, shape.max_len, bases.ctypes.data, exc_cap, C.byref(n_exc),
                                               quals.ctypes.data if quals is not None else None, names.ctypes.data, name_off.ctypes.data)
                if rc == 0:
                    out.append(PackedBatch(shape, bases, n_exc.value, quals, names, name_off))
                    continue
                if rc != 1:
                    raise ValueError("fem_seqfile_fill_packed failed (%d)" % rc)
            rc = L.fem_seqfile_fill(f, plan, threads, bases.ctypes.data, off.ctypes.data,
                                    quals.ctypes.data if quals is not None else None, names.ctypes.data, name_off.ctypes.data)
            if rc != 0:
                raise ValueError("fem_seqfile_fill failed (%d)" % rc)
            out.append(PlannedBatch(shape, bases, off, quals, names, name_off))
    finally:
        L.fem_seqfile_close(f)
    return out


def index_save(path, k, step, lookup, occ):
    lookup = np.ascontiguousarray(lookup, np.uint32)
    occ = np.ascontiguousarray(occ, np.uint64)
    rc = lib().fem_index_save(path.encode(), k, step, lookup.ctypes.data, len(occ), occ.ctypes.data)
    if rc != 0:
        raise OSError("fem_index_save failed (%d)" % rc)


def index_load(path):
    L = lib()
    k, step, n = C.c_int32(), C.c_int32(), C.c_uint64()
    lp, op = C.c_void_p(), C.c_void_p()
    rc = L.fem_index_load(path.encode(), C.byref(k), C.byref(step), C.byref(lp), C.byref(n), C.byref(op))
    if rc != 0:
        raise OSError("fem_index_load failed (%d)" % rc)
    lookup = _copy(lp.value, (1 << (2 * k.value)) + 1, np.uint32)
    occ = _copy(op.value, n.value, np.uint64)
    L.free(lp)
    L.free(op)
    return k.value, step.value, lookup, occ


# ------------------------------------------------------------------------------------------ mapping tail
class TailReference:
    def __init__(self, text, off, lens, names=None):
        self.text = np.ascontiguousarray(text, np.uint8)
        self.off = np.ascontiguousarray(off, np.uint64)
        self.len = np.ascontiguousarray(lens, np.uint32)
        names = names or ["chr%d" % (i + 1) for i in range(len(self.len))]
        raw = [n.encode() for n in names]
        self.name_off = np.zeros(len(raw) + 1, np.uint64)
        self.name_off[1:] = np.cumsum([len(r) for r in raw])
        self.names = np.frombuffer(b"".join(raw) + b"\0", np.uint8).copy()
        self.c = TailRef(self.text.ctypes.data, self.off.ctypes.data, self.len.ctypes.data, len(self.len),
                         self.names.ctypes.data, self.name_off.ctypes.data)


class RecordArrays:
    def __init__(self, r, n_reads):
        nr = int(r.n_records)
        self.rec_off = _copy(r.rec_off, n_reads + 1, np.uint64)
        self.flag = _copy(r.flag, nr, np.uint16)
        self.tid = _copy(r.tid, nr, np.uint32)
        self.pos0 = _copy(r.pos0, nr, np.uint32)
        self.nm = _copy(r.nm, nr, np.uint8)
        self.cigar_off = _copy(r.cigar_off, nr + 1, np.uint64)
        self.cigar = _copy(r.cigar, int(self.cigar_off[-1]), np.uint32)
        self.md_off = _copy(r.md_off, nr + 1, np.uint64)
        self.md = _copy(r.md, int(self.md_off[-1]), np.uint8)


def _tail_input(n_reads, cand_begin, cand_count, cand, ed, end):
    arrs = (np.ascontiguousarray(cand_begin, np.uint32), np.ascontiguousarray(cand_count, np.uint32),
            np.ascontiguousarray(cand, np.uint64), np.ascontiguousarray(ed, np.uint8),
            np.ascontiguousarray(end, np.int16))
    t = TailInput(n_reads, *[a.ctypes.data for a in arrs])
    return t, arrs


def tail_records(e, ref, read_bases, read_off, cand_begin, cand_count, cand, ed, end, threads=1):
    n_reads = len(read_off) - 1
    t, keep = _tail_input(n_reads, cand_begin, cand_count, cand, ed, end)
    bases = np.ascontiguousarray(read_bases, np.uint8)
    off = np.ascontiguousarray(read_off, np.uint64)
    r = Records()
    rc = lib().fem_tail_records(e, C.byref(ref.c), bases.ctypes.data, off.ctypes.data, C.byref(t), threads, C.byref(r))
    if rc != 0:
        raise RuntimeError("fem_tail_records failed (%d)" % rc)
    try:
        return RecordArrays(r, n_reads)
    finally:
        lib().fem_records_free(C.byref(r))


def records_sam(ref, names, read_bases, read_off, quals, rec, threads=1, parts=False):
    """SAM text for records already computed (fem_records_sam, or the no-copy fem_records_sam_parts form).  `rec` has
    rec_off/rec_begin, flag, tid, pos0, nm, cigar_off, cigar, md_off, md (RecordArrays or fem_amd.device.BatchRecords)."""
    n_reads = len(read_off) - 1
    bases = np.ascontiguousarray(read_bases, np.uint8)
    off = np.ascontiguousarray(read_off, np.uint64)
    q = np.ascontiguousarray(quals, np.uint8)
    raw = [n.encode() for n in names]
    name_off = np.zeros(n_reads + 1, np.uint64)
    name_off[1:] = np.cumsum([len(r) for r in raw])
    nm = np.frombuffer(b"".join(raw) + b"\0", np.uint8).copy()
    s = SeqSet(n_reads, bases.ctypes.data, off.ctypes.data, q.ctypes.data, nm.ctypes.data, name_off.ctypes.data)
    rb = np.ascontiguousarray(getattr(rec, "rec_begin", getattr(rec, "rec_off", None)), np.uint32)
    arrs = [rb, np.ascontiguousarray(rec.flag, np.uint16), np.ascontiguousarray(rec.tid, np.uint32),
            np.ascontiguousarray(rec.pos0, np.uint32), np.ascontiguousarray(rec.nm, np.uint8),
            np.ascontiguousarray(rec.cigar_off, np.uint32), np.ascontiguousarray(rec.cigar, np.uint32),
            np.ascontiguousarray(rec.md_off, np.uint32), np.ascontiguousarray(rec.md, np.uint8)]
    rv = RecordView(n_reads, len(arrs[1]), *[a.ctypes.data for a in arrs])
    L = lib()
    if not parts:
        p, n = C.c_void_p(), C.c_uint64()
        rc = L.fem_records_sam(C.byref(ref.c), C.byref(s), C.byref(rv), threads, C.byref(p), C.byref(n))
        if rc != 0:
            raise RuntimeError("fem_records_sam failed (%d)" % rc)
        text = _copy(p.value, n.value, np.uint8).tobytes().decode()
        L.free(p)
        return text
    buf, cap, na = C.c_void_p(), C.c_uint64(0), C.c_uint64(0)
    pt = (TextPart * threads)()
    rc = L.fem_records_sam_parts(C.byref(ref.c), C.byref(s), C.byref(rv), threads, C.byref(buf), C.byref(cap), pt, C.byref(na))
    if rc != 0:
        raise RuntimeError("fem_records_sam_parts failed (%d)" % rc)
    whole = _copy(buf.value, cap.value, np.uint8)
    text = b"".join(whole[int(x.offset):int(x.offset + x.length)].tobytes() for x in pt).decode()
    L.free(buf)
    return text, int(na.value)


def sam_header(ref):
    p, n = C.c_void_p(), C.c_uint64()
    lib().fem_sam_header(C.byref(ref.c), C.byref(p), C.byref(n))
    s = _copy(p.value, n.value, np.uint8).tobytes().decode()
    lib().free(p)
    return s


def tail_sam(e, ref, names, read_bases, read_off, quals, cand_begin, cand_count, cand, ed, end, threads=1):
    n_reads = len(read_off) - 1
    t, keep = _tail_input(n_reads, cand_begin, cand_count, cand, ed, end)
    bases = np.ascontiguousarray(read_bases, np.uint8)
    off = np.ascontiguousarray(read_off, np.uint64)
    q = np.ascontiguousarray(quals, np.uint8)
    raw = [n.encode() for n in names]
    name_off = np.zeros(n_reads + 1, np.uint64)
    name_off[1:] = np.cumsum([len(r) for r in raw])
    nm = np.frombuffer(b"".join(raw) + b"\0", np.uint8).copy()
    s = SeqSet(n_reads, bases.ctypes.data, off.ctypes.data, q.ctypes.data, nm.ctypes.data, name_off.ctypes.data)
    p, n = C.c_void_p(), C.c_uint64()
    rc = lib().fem_tail_sam(e, C.byref(ref.c), C.byref(s), C.byref(t), threads, C.byref(p), C.byref(n))
    if rc != 0:
        raise RuntimeError("fem_tail_sam failed (%d)" % rc)
    text = _copy(p.value, n.value, np.uint8).tobytes().decode()
    lib().free(p)
    return text
