import cProfile, pstats, sys, io, runpy
sys.argv = ["bench.py", "--steps", "300", "--warmup", "5", "--cpu-seconds", "0"]
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path("/root/repo/bench.py", run_name="__main__")
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
ps = pstats.Stats(pr, stream=s).sort_stats("tottime")
ps.print_stats(45)
print(s.getvalue())
