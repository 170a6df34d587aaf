import sys, os, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
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
for it in range(3):
    eng.reset_sample(); eng.submit_packed_device(packed.data_ptr(), qrows.data_ptr(), lens.data_ptr(), n, wpr, qs); eng.synchronize()
eng.set_profiling(True); eng.reset_kernel_time()
for it in range(10):
    eng.reset_sample(); eng.submit_packed_device(packed.data_ptr(), qrows.data_ptr(), lens.data_ptr(), n, wpr, qs)
eng.synchronize()
s = eng.stats()
t, c = eng.kernel_time(os.environ.get('KERNEL', 'sieve'))
print(os.environ.get('MLST_LIB', 'default'), os.environ.get('KERNEL', 'sieve') + ' ms/launch', round(t / c, 4), 'candidates', int(s.counters[3]))
