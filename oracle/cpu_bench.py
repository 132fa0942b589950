"""One oracle process of bench.py's cpu_baseline leg: loads the index, aligns its slice of the sample file with the CPU
oracle on ONE thread and prints {"reads", "seconds"} (the time excludes loading).  Test infrastructure, like all of oracle/:
    python -m oracle.cpu_bench <index prefix> <sample.bin> <read_len> <first read> <n reads> <first ordinal>
"""
import json
import sys
import time

import numpy as np


def main():
    prefix, sample, read_len, lo, n, ordinal = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
    from oracle import orc
    idx = orc.Index.load(prefix)
    reads = np.fromfile(sample, dtype=np.uint8, count=n * read_len, offset=lo * read_len)
    offs = np.arange(n + 1, dtype=np.uint64) * np.uint64(read_len)
    t0 = time.time()
    orc.align_batch_flat(orc.default_opt(), idx, reads.tobytes(), offs, first_ordinal=ordinal)
    print(json.dumps(dict(reads=n, seconds=time.time() - t0)))


if __name__ == "__main__":
    main()
