import sys, tempfile
sys.path.insert(0,'/root/repo')
import __graft_entry__ as ge; ge.build()
from metamlst_amd import synth
from metamlst_amd.index import load_index
from metamlst_amd.engine import Engine
d=tempfile.mkdtemp()
db=synth.make_ecoli_db(d+'/e.db', alleles_per_locus=1430, n_profiles=50)
idx=load_index(d+'/e.db')
st=db.profiles['ecoli'][3]
g,_=synth.make_genome(db,'ecoli',st,size=1_000_000)
b,q=synth.sample_reads(g,400_000)
fb,fq,off=synth.flatten_reads(b,q)
eng=Engine(0); eng.load_reference(idx); eng.submit_reads(fb,fq,off); s=eng.stats()
print('GPU counters',list(map(int,s.counters)), 'index bytes', eng.index_bytes())
