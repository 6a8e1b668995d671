# round 3 final measurement: PMC passes (c4 in the timed, overlapped mode; c3; c5), search work counters, the bench lines kept under profiles/r03/
set -x
mkdir -p gpurun_out/r03 profiles/r03
for wl in c4 c3 c5; do
  bash tools/r03/pmc.sh $wl > gpurun_out/r03/pmc_$wl.log 2>&1 || { tail -20 gpurun_out/r03/pmc_$wl.log; exit 1; }
done
for wl in c3 c4 c5; do
  GORIO_AMD_LIB=$PWD/tools/variants/stats.so timeout -k 10 400 python tools/search_work.py $wl 20 --merge > profiles/r03/search_work_$wl.txt 2>&1 || { tail -20 profiles/r03/search_work_$wl.txt; exit 1; }
done
timeout -k 10 400 python bench.py --steps 40 --warmup 3 > profiles/r03/bench_c4.json 2> gpurun_out/r03/bench_c4.err || { tail -5 gpurun_out/r03/bench_c4.err; exit 1; }
for k in 1 2 3; do timeout -k 10 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-check --no-exhaustive 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['value']), round(d['gp_windows_per_s']))" >> profiles/r03/bench_c4_repeats.txt; done
timeout -k 10 400 python bench.py --steps 20 --warmup 3 --no-overlap --no-cpu-baseline > profiles/r03/bench_c4_no_overlap.json 2> gpurun_out/r03/bench_c4no.err || exit 1
timeout -k 10 400 python bench.py --steps 20 --warmup 3 --optimizer lm > profiles/r03/bench_c4_lm.json 2> gpurun_out/r03/bench_c4lm.err || exit 1
timeout -k 10 400 python bench.py --workload c3 --steps 40 --warmup 3 > profiles/r03/bench_c3.json 2> gpurun_out/r03/bench_c3.err || { tail -5 gpurun_out/r03/bench_c3.err; exit 1; }
timeout -k 10 600 python bench.py --workload c5 --steps 10 --warmup 2 > profiles/r03/bench_c5.json 2> gpurun_out/r03/bench_c5.err || { tail -5 gpurun_out/r03/bench_c5.err; exit 1; }
timeout -k 10 900 python bench.py --latency > profiles/r03/latency.json 2> gpurun_out/r03/latency.err || { tail -5 gpurun_out/r03/latency.err; exit 1; }
timeout -k 10 1000 python -m pytest tests -m gpu -q > profiles/r03/gpu_tests_r03.txt 2>&1; tail -3 profiles/r03/gpu_tests_r03.txt
cp profiles/kernel_counters.json gpurun_out/r03/kernel_counters_final.json
rm -rf gpurun_out/r03/profiles_r03; cp -r profiles/r03 gpurun_out/r03/profiles_r03
python - <<'PY'
import json
for wl in ("c4","c4_no_overlap","c4_lm","c3","c5"):
    d=json.load(open(f"profiles/r03/bench_{wl}.json"))
    r=d["roofline"]
    print(wl, "value", round(d["value"]), "ms/step", round(d["ms_per_step"],3), "gp", d.get("gp_windows_per_s"), "| dom", r["kernel"], "frac", r["frac"], "useful", r.get("useful_frac"), "slow share", r.get("slow_instruction_share"), "| cpu", (d.get("cpu_baseline") or {}).get("value"))
    print("   ", {k:round(v,3) for k,v in d["device_ms_per_step"].items()})
print(open("profiles/r03/bench_c4_repeats.txt").read())
l=json.load(open("profiles/r03/latency.json"))
for e in l["latency"]: print({k:(round(v,3) if isinstance(v,float) else v) for k,v in e.items()})
PY
