"""Makes tests/golden/klib_sort.npz and klib_kseq.npz from the reference's own klib code (oracle/_ref/libfemref_klib.so =
/root/reference/src/kseq.h + ksort.h behind oracle/ref_klib.c).  Run in a container that has /root/reference:

    make -C oracle ref && python tests/golden/make_klib_golden.py

The vectors are inputs + what the reference code returned for them; no reference source is stored."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_klib  # noqa: E402
from tests import test_ref_klib as T  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    out = {}
    cases = [k for k in T._sort_cases() if len(k) <= 1000]
    for i, keys in enumerate(cases):
        _, perm = ref_klib.radix_sort(keys)
        out["keys_%d" % i], out["perm_%d" % i] = keys, perm
    out["n_cases"] = np.int64(len(cases))
    np.savez_compressed(os.path.join(HERE, "klib_sort.npz"), **out)

    out = {}
    rng = np.random.default_rng(4)
    files = [T.HAND_MADE[k] for k in sorted(T.HAND_MADE)] + [T._random_file(rng) for _ in range(60)]
    with tempfile.TemporaryDirectory() as d:
        for i, data in enumerate(files):
            p = os.path.join(d, "f.fq")
            with open(p, "wb") as f:
                f.write(data)
            recs, fatal = T.loader_view(*ref_klib.kseq_records(p))
            out["file_%d" % i] = np.frombuffer(data, np.uint8)
            out["fatal_%d" % i] = np.bool_(fatal)
            out["names_%d" % i] = np.frombuffer(b"".join(r[0] + b"\x00" for r in recs), np.uint8)
            out["seqs_%d" % i] = np.frombuffer(b"".join(r[2] + b"\x00" for r in recs), np.uint8)
    out["n_cases"] = np.int64(len(files))
    np.savez_compressed(os.path.join(HERE, "klib_kseq.npz"), **out)
    print("wrote klib_sort.npz (%d cases) and klib_kseq.npz (%d files)" % (len(cases), len(files)))


if __name__ == "__main__":
    main()
