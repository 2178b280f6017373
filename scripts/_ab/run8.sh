cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05a
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-launch-mix 2>gpurun_out/r05a/bench_census.err | tail -1 > gpurun_out/r05a/bench_census.json
tail -5 gpurun_out/r05a/bench_census.err
python - <<'PY'
import json
r=json.load(open("gpurun_out/r05a/bench_census.json"))
print(r["ms_per_step"])
rf=r["roofline"]
print({k:v for k,v in rf.items() if k not in ("kernels","families","dense_dgrad_probe","step","weakest_large_family")})
for k in rf.get("kernels",[]): print(k)
for k in rf.get("families",[]): print(k)
PY
