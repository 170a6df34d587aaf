cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp_ab
for rep in 1 2; do
for v in c_0bb6af7 head; do
  if [ $v = head ]; then unset MLST_LIB; else export MLST_LIB=$GRAFT_REPO_ROOT/build_variants/$v.so; fi
  python3 bench.py --cpu-seconds 0 --no-secondary > gpurun_out/exp_ab/${v}_$rep.json 2> gpurun_out/exp_ab/${v}_$rep.err
  python3 - <<PY
import json
j=json.loads(open("gpurun_out/exp_ab/${v}_$rep.json").read().strip().splitlines()[-1])
k=j["kernel_ms_per_launch_isolated"]
print("$v $rep", j["value"], j["ms_per_step"], j["serial_ms_per_step"], k["sieve_route"], k["sieve_probe"], k["extend"], k["seed"], k["accumulate"], k["pileup"], j["concordance"]["species_typed_correctly"])
PY
done
done
