"""Rate of the row-counting pass (feed.count_rows_in_range) over a page-cached feature TSV by thread count."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from deepsignal_plant_amd import feed
from tests.helpers import GOLDEN
data = open(os.path.join(GOLDEN, "f2_rows.tsv"), "rb").read()
path = "/tmp/cnt.tsv"
with open(path, "wb") as f:
    for _ in range(10):
        f.write(data * 1000)          # 4.2 GB
size = os.path.getsize(path)
for nt in (1, 1, 2, 4, 8, 16, 32):
    t = time.time(); n = feed.count_rows_in_range(path, 0, size, nthreads=nt); dt = time.time() - t
    print("%2d threads: %d rows %.2f s %.1f GB/s" % (nt, n, dt, size / dt / 1e9), flush=True)
os.remove(path)
