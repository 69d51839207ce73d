"""Rate of writing one large buffer to a file in /tmp: one write() against N threads of pwrite() on disjoint ranges
(what population.vcf.gz's writer can expect from the page cache of the box)."""
import os, sys, time, threading
import numpy as np

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 4 << 30
path = sys.argv[2] if len(sys.argv) > 2 else "/tmp/write_rate.bin"
chunk = int(float(sys.argv[3])) if len(sys.argv) > 3 else 1 << 30  # bytes per write() of the one-thread case
buf = np.random.default_rng(1).integers(0, 256, n, dtype=np.uint8)
mv = memoryview(buf)
for threads in (1, 2, 4, 8, 16):
    fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    t = time.time()
    if threads == 1:
        off = 0
        while off < n:
            off += os.write(fd, mv[off:off + chunk])
    else:
        step = (n + threads - 1) // threads
        def run(k):
            lo, hi = k * step, min(n, (k + 1) * step)
            while lo < hi:
                lo += os.pwrite(fd, mv[lo:min(hi, lo + (256 << 20))], lo)
        ths = [threading.Thread(target=run, args=(k,)) for k in range(threads)]
        [x.start() for x in ths]; [x.join() for x in ths]
    dt = time.time() - t
    os.close(fd)
    print("%2d thread(s)%s: %.2f GB/s" % (threads, " of %d MB writes" % (chunk >> 20) if threads == 1 else "", n / dt / 1e9), flush=True)
    os.remove(path)
