#!/bin/bash
# One GPU session (run from the repo root through gpurun), assembled from steps given on the command line:
#   tests [pytest args]     parity tier (default: the whole `-m gpu` tier)
#   bench <cfg>...          one bench line per configuration (headline only, 20 steps): ms/step + per-kernel times
#   prof <tag> <cfg>        rocprofv3 --kernel-trace --stats of bench.py --config <cfg>  -> gpurun_out/prof_<tag>_<cfg>
#   pmc <tag> <cfg>         PMC passes (each its own rocprofv3 run: FETCH_SIZE, WRITE_SIZE, two SQ groups)
#                           -> gpurun_out/pmc_<tag><cfg>_summary.txt  (tools/pmc_summary.py)
#   ab <tag> <cfg> [K=V..]  one bench line of <cfg> (40 steps) under the given environment (EDT_HIP_DEBUG_MODE=..., EDT_HIP_LIB=a variant
#                           build, EDT_BENCH_ALTERNATE=1 ...) -> gpurun_out/ab_<tag>.json: the A/B runs of round 5
#   final                   what the driver runs at round end: whole GPU tier, smoke(), the default bench line
#   fuzz <n>                randomised parity in the four tile-choice / pass-X modes
# Steps are separated by `--`, e.g.
#   gpurun -- ./tools/gpu_session.sh tests tests/test_gpu_parity.py -- bench cfg3 cfg3L -- prof r03 cfg3L -- pmc r03 cfg3L
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
mkdir -p gpurun_out
run_step() {
  local what=$1; shift
  case $what in
    tests)
      if [ $# -eq 0 ]; then set -- tests; fi
      python -m pytest "$@" -m gpu -x -q 2>&1 | tail -6 | tee -a gpurun_out/pytest_gpu.log ;;
    bench)
      for c in "$@"; do
        python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --config $c > gpurun_out/bench_$c.json 2> gpurun_out/bench_$c.err
        python - $c <<'PY'
import json, sys
c = sys.argv[1]
try:
    d = json.load(open(f"gpurun_out/bench_{c}.json"))
    print(c, d["ms_per_step"], d["roofline"]["kernel_ms"], "frac32B", d["roofline"]["whole_job_frac"], d["config"]["output_verified"])
except Exception as e:
    print(c, "ERR", e, open(f"gpurun_out/bench_{c}.err").read()[-600:])
PY
      done ;;
    ab)  # ab <tag> <cfg> [KEY=VALUE ...]
      local tag=$1 cfg=$2; shift 2
      env "$@" python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary --config $cfg > gpurun_out/ab_${tag}.json 2> gpurun_out/ab_${tag}.err
      python - $tag <<'PY'
import json, sys
t = sys.argv[1]
try:
    d = json.load(open(f"gpurun_out/ab_{t}.json"))
    print(t, d["ms_per_step"], d["roofline"]["kernel_ms"], "frac32B", d["roofline"]["whole_job_frac"], d["config"]["output_verified"])
except Exception as e:
    print(t, "ERR", e, open(f"gpurun_out/ab_{t}.err").read()[-800:])
PY
      ;;
    benchsz)  # benchsz <size> <cfg>...: the same at another edge length (cfg4 at 1024)
      local sz=$1; shift
      for c in "$@"; do
        python bench.py --steps 10 --warmup 2 --size $sz --no-cpu-baseline --no-secondary --config $c > gpurun_out/bench_${c}_$sz.json 2> gpurun_out/bench_${c}_$sz.err
        python - $c $sz <<'PY'
import json, sys
c, sz = sys.argv[1], sys.argv[2]
try:
    d = json.load(open(f"gpurun_out/bench_{c}_{sz}.json"))
    print(c, sz, d["ms_per_step"], d["roofline"]["kernel_ms"], "frac32B", d["roofline"]["whole_job_frac"], d["config"]["output_verified"])
except Exception as e:
    print(c, sz, "ERR", e, open(f"gpurun_out/bench_{c}_{sz}.err").read()[-600:])
PY
      done ;;
    prof)
      rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$1_$2 -o p -- python bench.py --steps ${BENCH_STEPS:-20} --warmup 3 --size ${BENCH_SIZE:-512} --no-cpu-baseline --no-secondary --config $2 > gpurun_out/prof_$1_$2.log 2>&1
      echo "prof $1 $2 rc=$?"
      f=$(find gpurun_out/prof_$1_$2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" ;;
    pmc)
      local i=0
      for ctrs in "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE GRBM_GUI_ACTIVE" \
                  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
                  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAIT_INST_LDS"; do
        i=$((i+1))
        if [ -n "$PMC_PASSES" ] && [ $i -gt $PMC_PASSES ]; then break; fi
        rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d gpurun_out/pmc_$1$2_$i -o p -- python bench.py --steps 3 --warmup 1 --size ${BENCH_SIZE:-512} --no-cpu-baseline --no-secondary --config $2 > gpurun_out/pmc_$1$2_$i.log 2>&1
        echo "pmc $1 $2 pass $i rc=$?"
      done
      python tools/pmc_summary.py $1$2 > gpurun_out/pmc_$1$2_summary.txt
      grep -A20 "k_column_pass_wave" gpurun_out/pmc_$1$2_summary.txt | grep -E "k_column|FETCH|WRITE|INSTS_VALU|WAVE_CYCLES|WAIT_INST_ANY" | head -16 ;;
    profshard)  # profshard <tag>: kernel stats of the N > 1 leg as a 1-rank RCCL dry run (4 chunks) -> gpurun_out/prof_<tag>_shard
      EDT_SHARD_CHUNKS=4 EDT_BENCH_FORCE_SHARDED=1 EDT_BENCH_VERIFY=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 \
        rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$1_shard -o p -- python bench.py --steps 20 --warmup 3 --no-selftest > gpurun_out/prof_$1_shard.log 2>&1
      echo "profshard $1 rc=$?"
      f=$(find gpurun_out/prof_$1_shard -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -9 "$f" ;;
    cmd)  # cmd <tag> <name> -- <command...>: kernel stats of any command -> gpurun_out/prof_<tag>_<name>
      local tag=$1 name=$2; shift 2
      rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_${name} -o p -- "$@" > gpurun_out/prof_${tag}_${name}.log 2>&1
      echo "cmd $tag $name rc=$?"
      f=$(find gpurun_out/prof_${tag}_${name} -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" ;;
    pmccmd)  # pmccmd <tag> <name> <command...>: FETCH_SIZE / WRITE_SIZE passes of any command -> gpurun_out/pmc_<tag><name>_{1,2}
      local tag=$1 name=$2; shift 2
      local i=0
      for ctrs in "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE GRBM_GUI_ACTIVE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS"; do
        i=$((i+1))
        rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d gpurun_out/pmc_${tag}${name}_$i -o p -- "$@" > gpurun_out/pmc_${tag}${name}_$i.log 2>&1
        echo "pmccmd $tag $name pass $i rc=$?"
      done
      python tools/pmc_summary.py ${tag}${name} > gpurun_out/pmc_${tag}${name}_summary.txt ;;
    final)
      python -m pytest tests -m gpu -x -q 2>&1 | tail -3
      python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
      python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
      python - <<'PY'
import json
d = json.load(open("gpurun_out/final_bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel"], d["roofline"]["whole_job_frac"], d["config"]["output_verified"], d["cpu_baseline"]["value"])
for s in d.get("secondary", []):
    print(s["config"], s.get("ms_per_step", s.get("gpu_seconds_total")), s.get("kernel_ms"), s.get("whole_job_frac"), s.get("output_verified"), s.get("error"))
PY
      ;;
    fuzz)
      local n=${1:-800}
      python tools/fuzz_gpu.py $n 61 2>&1 | tail -1
      EDT_HIP_DEBUG_MODE=0x4000 python tools/fuzz_gpu.py $((n / 2)) 62 2>&1 | tail -1
      EDT_HIP_DEBUG_MODE=0xC000 python tools/fuzz_gpu.py $((n / 2)) 63 2>&1 | tail -1
      EDT_HIP_DEBUG_MODE=0x2000 python tools/fuzz_gpu.py $((n / 2)) 64 2>&1 | tail -1
      EDT_HIP_DEBUG_MODE=0x100000 python tools/fuzz_gpu.py $((n / 2)) 65 2>&1 | tail -1
      EDT_HIP_DEBUG_MODE=0x400000 python tools/fuzz_gpu.py $n 67 2>&1 | tail -1
      FUZZ_MAX_AXIS=2100 python tools/fuzz_gpu.py $((n / 2)) 66 2>&1 | tail -1 ;;
    run) "$@" ;;
    *) echo "unknown step $what" ;;
  esac
}
args=()
for a in "$@"; do
  if [ "$a" == "--" ]; then run_step "${args[@]}"; args=(); else args+=("$a"); fi
done
[ ${#args[@]} -gt 0 ] && run_step "${args[@]}"
