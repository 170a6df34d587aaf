"""Host time of the individual engine calls of one step (are the asynchronous ones asynchronous?)."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from metamlst_amd import synth
from metamlst_amd.index import load_index
from metamlst_amd.engine import Engine
d = tempfile.mkdtemp()
db = synth.make_ecoli_db(d + '/e.db', alleles_per_locus=1430, n_profiles=50)
idx = load_index(d + '/e.db')
g, _ = synth.make_genome(db, 'ecoli', db.profiles['ecoli'][3], size=4_600_000)
dev = torch.device('cuda', 0)
eng = Engine(0); eng.load_reference(idx)
n = 10_000_000
packed, qrows, lens, wpr, qs = bench.synth_reads_gpu(eng, torch, dev, g, n, 150, seed=5)
T = {}
def tm(name, f, *a, **k):
    t = time.perf_counter(); r = f(*a, **k); T.setdefault(name, []).append((time.perf_counter() - t) * 1e6); return r
for it in range(30):
    tm('reset_sample', eng.reset_sample)
    tm('submit_packed_device', eng.submit_packed_device, packed.data_ptr(), qrows.data_ptr(), lens.data_ptr(), n, wpr, qs)
    tm('typing_enqueue', eng.typing_enqueue, 100)
    tm('typing_fetch', eng.typing_fetch)
for k, v in T.items():
    v = sorted(v[5:]); print('%-24s median %8.1f us' % (k, v[len(v) // 2]))
