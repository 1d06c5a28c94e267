"""ctypes binding of libfemhip.so (include/fem_hip.h).  One Device == one fem_dev handle == one GPU."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_HIP = None

#: every symbol include/fem_hip.h declares
ABI_SYMBOLS = [
    "fem_dev_open", "fem_dev_close", "fem_strerror", "fem_dev_last_error", "fem_dev_limits",
    "fem_dev_upload_index", "fem_dev_upload_reference", "fem_dev_build_index", "fem_dev_fetch_index",
    "fem_dev_map_batch_submit", "fem_dev_map_batch_wait",
    "fem_dev_stage_reads", "fem_dev_stage_info", "fem_dev_acquire_stage", "fem_dev_commit_stage", "fem_dev_commit_stage_uniform", "fem_dev_packed_layout", "fem_dev_commit_stage_packed", "fem_dev_map_staged", "fem_dev_sync", "fem_dev_fetch_stats", "fem_dev_fetch", "fem_dev_fetch_packed",
    "fem_dev_fetch_records", "fem_dev_seed_kernel", "fem_dev_index_info",
    "fem_dev_upload_reference_names", "fem_dev_acquire_text_stage", "fem_dev_commit_text_stage", "fem_dev_commit_names_stage", "fem_dev_sam_quals", "fem_dev_reserve_text", "fem_dev_reserve_batch", "fem_set_blocking_waits", "fem_dev_fetch_sam", "fem_dev_fetch_sam_nowait", "fem_dev_sam_wait",
    "fem_dev_set_timing", "fem_dev_reset_timing", "fem_dev_kernel_time", "fem_dev_copy_bandwidth",
    "fem_dev_h2d_bandwidth",
    "fem_device_numa", "fem_bind_thread_near_device",
    "fem_dev_allreduce_stats",
]


class FemError(RuntimeError):
    pass


class Params(C.Structure):
    """fem_params == FEMArgs of the reference (src/utils.h:63-70); map always uses k=12, step=3."""
    _fields_ = [("k", C.c_int32), ("step", C.c_int32), ("e", C.c_int32), ("a", C.c_int32)]


class _ReadBatch(C.Structure):
    _fields_ = [("bases", C.c_void_p), ("offsets", C.c_void_p), ("n_reads", C.c_uint64)]


class _BatchResult(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("n_candidates", C.c_uint64), ("cand_begin", C.c_void_p),
                ("cand_count", C.c_void_p), ("cand", C.c_void_p), ("ed", C.c_void_p), ("end", C.c_void_p),
                ("stats", C.c_uint64 * 5)]


class _BatchPacked(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("n_candidates", C.c_uint64), ("count", C.c_void_p), ("seg_begin", C.c_void_p),
                ("cand", C.c_void_p), ("ed", C.c_void_p), ("end", C.c_void_p), ("big", C.c_void_p), ("n_big", C.c_uint32),
                ("stats", C.c_uint64 * 5)]


class _BatchRecords(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("n_records", C.c_uint64), ("rec_begin", C.c_void_p), ("flag", C.c_void_p),
                ("tid", C.c_void_p), ("pos0", C.c_void_p), ("nm", C.c_void_p), ("cigar_off", C.c_void_p),
                ("cigar", C.c_void_p), ("md_off", C.c_void_p), ("md", C.c_void_p), ("stats", C.c_uint64 * 5)]


class _BatchSam(C.Structure):
    _fields_ = [("text", C.c_void_p), ("len", C.c_uint64), ("n_reads", C.c_uint64), ("n_records", C.c_uint64),
                ("n_asserted", C.c_uint64), ("stats", C.c_uint64 * 5)]


def hip_library_path():
    # FEM_HIP_LIBRARY: another build of the same library (ablation / tuning builds under gpurun_out/, measurement only)
    return os.environ.get("FEM_HIP_LIBRARY") or os.path.join(_HERE, "csrc", "libfemhip.so")


def load_hip():
    """Load libfemhip.so; raise loudly if it has not been built (no fallback of any kind)."""
    global _HIP
    if _HIP is not None:
        return _HIP
    path = hip_library_path()
    if not os.path.exists(path):
        raise FemError("libfemhip.so is missing (%s): run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "or `make -C fem_amd/csrc`" % path)
    L = C.CDLL(path)
    vp, u64, i32 = C.c_void_p, C.c_uint64, C.c_int32
    L.fem_dev_open.argtypes = [C.c_int, C.POINTER(vp)]
    L.fem_dev_close.argtypes = [vp]
    L.fem_strerror.restype = C.c_char_p
    L.fem_strerror.argtypes = [C.c_int]
    L.fem_dev_last_error.restype = C.c_char_p
    L.fem_dev_last_error.argtypes = [vp]
    L.fem_dev_limits.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(i32)]
    L.fem_dev_upload_index.argtypes = [vp, i32, i32, vp, u64, vp, u64]
    L.fem_dev_upload_reference.argtypes = [vp, C.c_uint32, C.POINTER(vp), vp]
    L.fem_dev_build_index.argtypes = [vp, i32, i32, vp, vp, u64, C.POINTER(u64)]
    L.fem_dev_fetch_index.argtypes = [vp, vp, vp, u64]
    L.fem_dev_map_batch_submit.argtypes = [vp, C.c_int, C.POINTER(Params), C.POINTER(_ReadBatch)]
    L.fem_dev_map_batch_wait.argtypes = [vp, C.c_int, C.POINTER(_BatchResult)]
    L.fem_dev_stage_reads.argtypes = [vp, C.c_int, C.POINTER(_ReadBatch)]
    L.fem_dev_acquire_stage.argtypes = [vp, C.c_int, u64, u64, C.POINTER(vp), C.POINTER(vp)]
    L.fem_dev_commit_stage.argtypes = [vp, C.c_int, u64, C.c_uint32]
    L.fem_dev_commit_stage_uniform.argtypes = [vp, C.c_int, u64, C.c_uint32]
    L.fem_dev_packed_layout.argtypes = [u64, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(u64), C.POINTER(u64)]
    L.fem_dev_commit_stage_packed.argtypes = [vp, C.c_int, u64, C.c_uint32, u64]
    L.fem_dev_map_staged.argtypes = [vp, C.c_int, C.POINTER(Params)]
    L.fem_dev_sync.argtypes = [vp, C.c_int]
    L.fem_dev_fetch_stats.argtypes = [vp, C.c_int, vp]
    L.fem_dev_fetch.argtypes = [vp, C.c_int, C.POINTER(_BatchResult)]
    L.fem_dev_fetch_packed.argtypes = [vp, C.c_int, C.POINTER(_BatchPacked)]
    L.fem_dev_fetch_records.argtypes = [vp, C.c_int, C.POINTER(_BatchRecords)]
    L.fem_dev_index_info.argtypes = [vp, C.c_char_p, u64]
    L.fem_dev_seed_kernel.restype = C.c_char_p
    L.fem_dev_seed_kernel.argtypes = [vp, C.POINTER(Params)]
    L.fem_dev_set_timing.argtypes = [vp, C.c_int]
    L.fem_dev_reset_timing.argtypes = [vp]
    L.fem_dev_kernel_time.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(u64)]
    L.fem_dev_copy_bandwidth.argtypes = [vp, u64, C.c_int, C.POINTER(C.c_double)]
    L.fem_dev_h2d_bandwidth.argtypes = [vp, u64, C.c_int, C.POINTER(C.c_double)]
    L.fem_dev_allreduce_stats.argtypes = [C.POINTER(vp), C.c_int, vp]
    L.fem_dev_upload_reference_names.argtypes = [vp, C.c_uint32, C.c_char_p, vp]
    L.fem_dev_acquire_text_stage.argtypes = [vp, C.c_int, u64, u64, u64, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.fem_dev_commit_text_stage.argtypes = [vp, C.c_int, u64, u64]
    if hasattr(L, "fem_dev_commit_names_stage"):
        L.fem_dev_commit_names_stage.argtypes = [vp, C.c_int, u64, u64]
        L.fem_dev_sam_quals.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(u64)]
    L.fem_dev_reserve_text.argtypes = [vp, C.c_int, u64, u64, u64, u64]
    if hasattr(L, "fem_dev_reserve_batch"):  # (FEM_HIP_LIBRARY may name an older build of the library: measurement only)
        L.fem_dev_reserve_batch.argtypes = [vp, C.c_int, u64, u64, C.c_uint32, C.POINTER(Params)]
    L.fem_dev_fetch_sam.argtypes = [vp, C.c_int, C.POINTER(_BatchSam)]
    L.fem_dev_fetch_sam_nowait.argtypes = [vp, C.c_int, C.POINTER(_BatchSam)]
    L.fem_dev_sam_wait.argtypes = [vp, C.c_int]
    L.fem_dev_stage_info.argtypes = [vp, C.c_int, C.POINTER(u64), C.POINTER(C.c_int32)]
    L.fem_device_numa.argtypes = [C.c_int, C.POINTER(C.c_int32), C.c_char_p, u64]
    L.fem_bind_thread_near_device.argtypes = [C.c_int]
    _HIP = L
    return L


def device_numa(device=0):
    """(NUMA node of the host memory next to GPU `device` or -1, that node's CPUs as a cpulist string)."""
    node = C.c_int32(-1)
    buf = C.create_string_buffer(4096)
    rc = load_hip().fem_device_numa(int(device), C.byref(node), buf, 4096)
    if rc != 0:
        raise FemError("fem_device_numa: %d" % rc)
    return node.value, buf.value.decode()


def bind_near_device(device=0):
    """Restricts the calling thread (and the threads it starts later) to the CPUs next to GPU `device`; True if bound."""
    return load_hip().fem_bind_thread_near_device(int(device)) == 0


def packed_layout(n_reads, read_len):
    """(bytes per read, offset of the exception positions behind the codes, most exceptions a packed batch may carry)
    of fem_dev_commit_stage_packed's staging layout."""
    bpr, off, cap = C.c_uint32(), C.c_uint64(), C.c_uint64()
    rc = load_hip().fem_dev_packed_layout(int(n_reads), int(read_len), C.byref(bpr), C.byref(off), C.byref(cap))
    if rc != 0:
        raise FemError("fem_dev_packed_layout: %d" % rc)
    return bpr.value, off.value, cap.value


def pack_reads(bases, n_reads, read_len, out):
    """numpy restatement of the packed form (tests): characters of n_reads reads of read_len -> codes + exceptions
    written into `out` (a staging view); returns the number of exceptions."""
    bpr, exc_off, exc_cap = packed_layout(n_reads, read_len)
    b = np.asarray(bases[:n_reads * read_len], dtype=np.uint8).reshape(n_reads, read_len)
    ok = (b == 65) | (b == 67) | (b == 71) | (b == 84)
    code = np.where(ok, ((b >> 1) ^ (b >> 2)) & 3, 0).astype(np.uint8)
    pad = bpr * 4 - read_len
    if pad:
        code = np.concatenate([code, np.zeros((n_reads, pad), np.uint8)], axis=1)
    q = code.reshape(n_reads, bpr, 4)
    out[:n_reads * bpr] = (q[:, :, 0] | (q[:, :, 1] << 2) | (q[:, :, 2] << 4) | (q[:, :, 3] << 6)).reshape(-1)
    out[n_reads * bpr:exc_off] = 0
    pos = np.flatnonzero(~ok.reshape(-1)).astype(np.uint32)
    n_exc = len(pos)
    if n_exc > exc_cap:
        raise FemError("too many exceptions for a packed batch")
    out[exc_off:exc_off + 4 * n_exc] = pos.view(np.uint8)
    out[exc_off + 4 * n_exc:exc_off + 5 * n_exc] = b.reshape(-1)[pos]
    return n_exc


def _copy(ptr, n, dtype, copy=True):
    n = int(n)
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    a = np.frombuffer(buf, dtype=dtype, count=n)
    return a.copy() if copy else a


class BatchResult:
    """fem_batch_result: host copies of the arrays, or (copy=False) views of the handle's pinned result buffers,
    valid until the slot is staged or mapped again."""

    def __init__(self, r, copy=True):
        n2 = 2 * int(r.n_reads)
        nc = int(r.n_candidates)
        self.n_reads = int(r.n_reads)
        self.n_candidates = nc
        self.cand_begin = _copy(r.cand_begin, n2, np.uint32, copy)
        self.cand_count = _copy(r.cand_count, n2, np.uint32, copy)
        self.cand = _copy(r.cand, nc, np.uint64, copy)
        self.ed = _copy(r.ed, nc, np.uint8, copy)
        self.end = _copy(r.end, nc, np.int16, copy)
        self.stats = np.array(list(r.stats), dtype=np.uint64)

    def per_strand(self):
        """Candidates regrouped in (read, strand) order: offsets[2n+1], cand, ed, end — the layout the oracle uses."""
        cnt = self.cand_count.astype(np.int64)
        off = np.zeros(len(cnt) + 1, dtype=np.uint64)
        off[1:] = np.cumsum(cnt)
        total = int(off[-1])
        idx = np.zeros(total, dtype=np.int64)
        if total:
            starts = np.repeat(self.cand_begin.astype(np.int64) - off[:-1].astype(np.int64), cnt)
            idx = np.arange(total, dtype=np.int64) + starts
        return off, self.cand[idx], self.ed[idx], self.end[idx]


class BatchPacked:
    """fem_batch_packed (include/fem_hip.h): the outcome in the form that crosses the link — one byte per strand, one offset
    per 256 strands, the candidates without padding.  copy=False: views of the handle's pinned buffers."""

    def __init__(self, r, copy=True):
        n2 = 2 * int(r.n_reads)
        nc = int(r.n_candidates)
        self.n_reads = int(r.n_reads)
        self.n_candidates = nc
        self.count = _copy(r.count, n2, np.uint8, copy)
        self.seg_begin = _copy(r.seg_begin, (n2 + 255) // 256, np.uint32, copy)
        self.cand = _copy(r.cand, nc, np.uint64, copy)
        self.ed = _copy(r.ed, nc, np.uint8, copy)
        self.end = _copy(r.end, nc, np.int16, copy)
        self.n_big = int(r.n_big)
        self.big = _copy(r.big, 2 * self.n_big, np.uint32, True).reshape(-1, 2)
        self.stats = np.array(list(r.stats), dtype=np.uint64)
        self.d2h_bytes = n2 + 4 * len(self.seg_begin) + 11 * nc + 8 * self.n_big

    def counts(self):
        """Candidates per strand, the strands listed in big[] with their real counts."""
        cnt = self.count.astype(np.int64)
        for s_, c_ in self.big:
            cnt[int(s_)] = int(c_)
        return cnt

    def per_strand(self):
        """Candidates regrouped in (read, strand) order: offsets[2n+1], cand, ed, end — the layout the oracle uses."""
        cnt = self.counts()
        off = np.zeros(len(cnt) + 1, dtype=np.uint64)
        off[1:] = np.cumsum(cnt)
        total = int(off[-1])
        idx = np.zeros(total, dtype=np.int64)
        if total:
            n_seg = len(self.seg_begin)
            pad = np.zeros(n_seg * 256, dtype=np.int64)
            pad[:len(cnt)] = cnt
            within = np.cumsum(pad.reshape(n_seg, 256), axis=1) - pad.reshape(n_seg, 256)  # exclusive prefix inside a segment
            begin = (self.seg_begin.astype(np.int64)[:, None] + within).reshape(-1)[:len(cnt)]
            idx = np.arange(total, dtype=np.int64) + np.repeat(begin - off[:-1].astype(np.int64), cnt)
        return off, self.cand[idx], self.ed[idx], self.end[idx]


class BatchRecords:
    """Host copy of fem_batch_records: the batch's output records in the reference's order."""

    def __init__(self, r):
        n, nr = int(r.n_reads), int(r.n_records)
        self.n_reads, self.n_records = n, nr
        self.rec_begin = _copy(r.rec_begin, n + 1, np.uint32)
        self.flag = _copy(r.flag, nr, np.uint16)
        self.tid = _copy(r.tid, nr, np.uint32)
        self.pos0 = _copy(r.pos0, nr, np.uint32)
        self.nm = _copy(r.nm, nr, np.uint8)
        self.cigar_off = _copy(r.cigar_off, nr + 1, np.uint32)
        self.cigar = _copy(r.cigar, int(self.cigar_off[-1]) if nr + 1 else 0, np.uint32)
        self.md_off = _copy(r.md_off, nr + 1, np.uint32)
        self.md = _copy(r.md, int(self.md_off[-1]) if nr + 1 else 0, np.uint8)
        self.stats = np.array(list(r.stats), dtype=np.uint64)


class Device:
    """One GPU: resident index + reference, batches mapped through slots."""

    def __init__(self, device=0):
        self._L = load_hip()
        h = C.c_void_p()
        rc = self._L.fem_dev_open(device, C.byref(h))
        if rc != 0:
            raise FemError("fem_dev_open(%d) failed: %s — the HIP path needs a GPU, there is no CPU fallback"
                           % (device, self._L.fem_strerror(rc).decode()))
        self._h = h
        self._keep = []

    def close(self):
        if self._h:
            self._L.fem_dev_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise FemError("%s: %s" % (self._L.fem_strerror(rc).decode(),
                                       self._L.fem_dev_last_error(self._h).decode()))

    def limits(self):
        m, s = C.c_uint32(), C.c_int32()
        self._check(self._L.fem_dev_limits(self._h, C.byref(m), C.byref(s)))
        return m.value, s.value

    def upload_index(self, k, step, lookup, occ, n_occ=None):
        lookup = np.ascontiguousarray(lookup, dtype=np.uint32)
        occ = np.ascontiguousarray(occ, dtype=np.uint64)
        n_occ = len(occ) if n_occ is None else n_occ
        self._check(self._L.fem_dev_upload_index(self._h, k, step, lookup.ctypes.data, len(lookup), occ.ctypes.data,
                                                 n_occ))

    def upload_reference(self, seqs):
        """seqs: list of bytes / uint8 arrays (raw FASTA characters)."""
        arrs = [np.frombuffer(s, dtype=np.uint8) if isinstance(s, (bytes, bytearray)) else np.ascontiguousarray(s, np.uint8)
                for s in seqs]
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        lens = np.array([len(a) for a in arrs], dtype=np.uint32)
        self._check(self._L.fem_dev_upload_reference(self._h, len(arrs), ptrs, lens.ctypes.data))

    def build_index(self, k=12, step=3, fetch=True):
        n = C.c_uint64()
        self._check(self._L.fem_dev_build_index(self._h, k, step, None, None, 0, C.byref(n)))
        if not fetch:
            return int(n.value), None, None
        lookup = np.zeros((1 << (2 * k)) + 1, dtype=np.uint32)
        occ = np.zeros(max(int(n.value), 1), dtype=np.uint64)
        self._check(self._L.fem_dev_fetch_index(self._h, lookup.ctypes.data, occ.ctypes.data, len(occ)))
        return int(n.value), lookup, occ[:n.value]

    @staticmethod
    def _batch(bases, offsets):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        return _ReadBatch(bases.ctypes.data, offsets.ctypes.data, len(offsets) - 1), (bases, offsets)

    def stage_reads(self, bases, offsets, slot=0):
        b, keep = self._batch(bases, offsets)
        self._check(self._L.fem_dev_stage_reads(self._h, slot, C.byref(b)))

    def stage_info(self, slot=0):
        """(bytes the slot's last staging sent over the link, True if as 2-bit codes)."""
        nb, pk = C.c_uint64(0), C.c_int32(0)
        self._check(self._L.fem_dev_stage_info(self._h, slot, C.byref(nb), C.byref(pk)))
        return nb.value, bool(pk.value)

    def acquire_stage(self, n_reads_cap, n_bases_cap, slot=0):
        """The slot's pinned staging buffers as numpy views (bases uint8[n_bases_cap + 64], offsets uint64[n_reads_cap + 1]):
        a parser (or a generator) writes the batch straight into them; commit_stage() then starts the H2D copy."""
        pb, po = C.c_void_p(), C.c_void_p()
        self._check(self._L.fem_dev_acquire_stage(self._h, slot, n_reads_cap, n_bases_cap, C.byref(pb), C.byref(po)))
        bases = np.frombuffer((C.c_char * (int(n_bases_cap) + 64)).from_address(pb.value), dtype=np.uint8)
        offs = np.frombuffer((C.c_char * (8 * (int(n_reads_cap) + 1))).from_address(po.value), dtype=np.uint64)
        return bases, offs

    def commit_stage(self, n_reads, max_len, slot=0, uniform=False):
        """uniform=True: every read has exactly max_len characters; the offsets are generated on the device."""
        if uniform:
            self._check(self._L.fem_dev_commit_stage_uniform(self._h, slot, n_reads, max_len))
        else:
            self._check(self._L.fem_dev_commit_stage(self._h, slot, n_reads, max_len))

    def commit_stage_packed(self, n_reads, read_len, n_exc=0, slot=0):
        """The slot's staging holds the batch at two bits per base + n_exc exceptions (packed_layout)."""
        self._check(self._L.fem_dev_commit_stage_packed(self._h, slot, n_reads, read_len, n_exc))

    def map_staged(self, e=3, a=1, k=12, step=3, slot=0):
        p = Params(k, step, e, a)
        self._check(self._L.fem_dev_map_staged(self._h, slot, C.byref(p)))

    def sync(self, slot=0):
        self._check(self._L.fem_dev_sync(self._h, slot))

    def fetch_stats(self, slot=0):
        st = np.zeros(5, dtype=np.uint64)
        self._check(self._L.fem_dev_fetch_stats(self._h, slot, st.ctypes.data))
        return st

    def fetch(self, slot=0, copy=True):
        r = _BatchResult()
        self._check(self._L.fem_dev_fetch(self._h, slot, C.byref(r)))
        return BatchResult(r, copy)

    def fetch_packed(self, slot=0, copy=True):
        """fem_dev_fetch_packed: the same outcome at a third of the bytes over the link."""
        r = _BatchPacked()
        self._check(self._L.fem_dev_fetch_packed(self._h, slot, C.byref(r)))
        return BatchPacked(r, copy)

    def seed_kernel(self, e=3, a=1, k=12, step=3):
        p = Params(k, step, e, a)
        return self._L.fem_dev_seed_kernel(self._h, C.byref(p)).decode()

    def index_info(self):
        """fem_dev_index_info: which derived tables the resident index has."""
        buf = C.create_string_buffer(512)
        self._check(self._L.fem_dev_index_info(self._h, buf, 512))
        return buf.value.decode()

    def reserve_batch(self, n_reads, n_records, max_len, e=3, a=1, k=12, step=3, slot=0):
        """fem_dev_reserve_batch: the slot's allocations for batches of this shape, made now."""
        p = Params(k, step, e, a)
        self._check(self._L.fem_dev_reserve_batch(self._h, slot, n_reads, n_records, max_len, C.byref(p)))

    def fetch_records(self, slot=0):
        """The device mapping tail: sorted records with CIGAR and MD (fem_dev_fetch_records)."""
        r = _BatchRecords()
        self._check(self._L.fem_dev_fetch_records(self._h, slot, C.byref(r)))
        return BatchRecords(r)

    def upload_reference_names(self, names):
        """names: list of str / bytes, one per reference sequence (the @SQ names)."""
        raw = [n.encode() if isinstance(n, str) else bytes(n) for n in names]
        off = np.zeros(len(raw) + 1, np.uint64)
        off[1:] = np.cumsum([len(n) for n in raw])
        self._check(self._L.fem_dev_upload_reference_names(self._h, len(raw), b"".join(raw), off.ctypes.data))

    def stage_text(self, quals, names, slot=0, quals_on_host=False):
        """Qualities (uint8 array / bytes, same offsets as the staged bases) and read names (list) of the slot's batch.
        quals_on_host: fem_dev_commit_names_stage — only the names go to the device, fetch_sam(quals=, offsets=) puts the
        qualities into the text."""
        raw = [n.encode() if isinstance(n, str) else bytes(n) for n in names]
        q = np.frombuffer(bytes(quals), np.uint8) if not isinstance(quals, np.ndarray) else quals
        nn = sum(len(n) for n in raw)
        pq, pn, po = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self._check(self._L.fem_dev_acquire_text_stage(self._h, slot, len(raw), len(q), nn, C.byref(pq), C.byref(pn), C.byref(po)))
        if len(q):
            C.memmove(pq.value, q.ctypes.data, len(q))
        if nn:
            C.memmove(pn.value, b"".join(raw), nn)
        off = np.zeros(len(raw) + 1, np.uint64)
        off[1:] = np.cumsum([len(n) for n in raw])
        C.memmove(po.value, off.ctypes.data, 8 * (len(raw) + 1))
        if quals_on_host:
            self._check(self._L.fem_dev_commit_names_stage(self._h, slot, len(raw), nn))
        else:
            self._check(self._L.fem_dev_commit_text_stage(self._h, slot, len(raw), nn))

    def fetch_sam(self, slot=0, nowait=False, quals=None, offsets=None):
        """(SAM text of the slot's batch as bytes, n_records, n_asserted, stats) — rendered on the device.
        nowait: through fem_dev_fetch_sam_nowait + fem_dev_sam_wait."""
        r = _BatchSam()
        if nowait:
            self._check(self._L.fem_dev_fetch_sam_nowait(self._h, slot, C.byref(r)))
            self._check(self._L.fem_dev_sam_wait(self._h, slot))
        else:
            self._check(self._L.fem_dev_fetch_sam(self._h, slot, C.byref(r)))
        if quals is not None and r.len:  # the batch went without its qualities: into the fields the device left open
            from fem_amd import host
            qa, nq = C.c_void_p(), C.c_uint64()
            self._check(self._L.fem_dev_sam_quals(self._h, slot, C.byref(qa), C.byref(nq)))
            q = np.ascontiguousarray(quals, dtype=np.uint8)
            o = np.ascontiguousarray(offsets, dtype=np.uint64)
            rc = host.lib().fem_sam_fill_quals(r.text, r.len, qa.value, nq.value, q.ctypes.data, o.ctypes.data, 0, 3)
            if rc:
                raise FemError("fem_sam_fill_quals failed (%d)" % rc)
        text = C.string_at(r.text, r.len) if r.len else b""
        return text, int(r.n_records), int(r.n_asserted), np.array(list(r.stats), dtype=np.uint64)

    def map_batch(self, bases, offsets, e=3, a=1, k=12, step=3, slot=0):
        b, keep = self._batch(bases, offsets)
        p = Params(k, step, e, a)
        self._check(self._L.fem_dev_map_batch_submit(self._h, slot, C.byref(p), C.byref(b)))
        r = _BatchResult()
        self._check(self._L.fem_dev_map_batch_wait(self._h, slot, C.byref(r)))
        return BatchResult(r)

    def set_timing(self, on=True):
        self._check(self._L.fem_dev_set_timing(self._h, int(on)))

    def reset_timing(self):
        self._check(self._L.fem_dev_reset_timing(self._h))

    def kernel_time(self, kernel):
        ms, n = C.c_double(), C.c_uint64()
        self._check(self._L.fem_dev_kernel_time(self._h, kernel, C.byref(ms), C.byref(n)))
        return ms.value, int(n.value)

    def h2d_bandwidth(self, nbytes=1 << 28, iters=8):
        g = C.c_double()
        self._check(self._L.fem_dev_h2d_bandwidth(self._h, nbytes, iters, C.byref(g)))
        return g.value

    def copy_bandwidth(self, nbytes=1 << 30, iters=10):
        g = C.c_double()
        self._check(self._L.fem_dev_copy_bandwidth(self._h, nbytes, iters, C.byref(g)))
        return g.value
