import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as ge; ge.build()
from metamlst_amd import synth
from metamlst_amd.index import load_index
from metamlst_amd.pipeline import make_engines
import tempfile
d = tempfile.mkdtemp()
sdb = synth.make_ecoli_db(d + "/e.db", alleles_per_locus=60, n_profiles=10)
idx = load_index(sdb.path)
engines = make_engines(idx, 0, 6)
for e in engines: e.synchronize()
for r in range(3):
    for k, e in enumerate(engines):
        t0 = time.perf_counter(); e.synchronize(); t1 = time.perf_counter(); e.set_cu_partition(k % 2, 2); t2 = time.perf_counter()
        print("run %d engine %d: sync %.2f ms, set_cu_partition %.2f ms" % (r, k, (t1 - t0) * 1e3, (t2 - t1) * 1e3))
    for e in engines: e.set_cu_partition(0, 1)
# the same from six threads at once (does the runtime make CU-masked streams side by side?)
from concurrent.futures import ThreadPoolExecutor
for r in range(3):
    t0 = time.perf_counter()
    with ThreadPoolExecutor(6) as ex:
        list(ex.map(lambda ke: ke[1].set_cu_partition(ke[0] % 2, 2), enumerate(engines)))
    t1 = time.perf_counter()
    print("run %d: six engines placed from six threads in %.2f ms" % (r, (t1 - t0) * 1e3))
    for e in engines: e.set_cu_partition(0, 1)
