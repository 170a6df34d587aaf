"""Wall-clock per call of the pieces of the host tail (typing.type_sample and what it calls) during a serial bench run."""
import os, sys, time, collections, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from metamlst_amd import typing as T, engine as E, merge as M, db as D
acc = collections.defaultdict(lambda: [0.0, 0])
def wrap(mod, name, label=None):
    f = getattr(mod, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); d = time.perf_counter() - t
        acc[label or name][0] += d; acc[label or name][1] += 1
        return r
    setattr(mod, name, g)
for n in ("_detected_loci", "pick_alleles_fast", "build_consensus_from_letters", "nfo_line", "type_sample"):
    wrap(T, n)
for n in ("consensus", "stats", "submit_packed_device", "reset_sample", "typing_enqueue", "typing_fetch", "set_read_index_base"):
    wrap(E.Engine, n, "Engine." + n)
wrap(M, "parse_nfo_line"); wrap(M.SpeciesSession, "add_sample", "SpeciesSession.add_sample")
wrap(D.DbCache, "sequenceFind", "DbCache.sequenceFind")
sys.argv = ["bench.py", "--steps", "200", "--warmup", "5", "--cpu-seconds", "0", "--pipeline", os.environ.get("PIPE", "1")]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
for k, (t, n) in sorted(acc.items(), key=lambda x: -x[1][0]):
    print("%-40s %6d calls  %8.1f us/call" % (k, n, t / n * 1e6))
