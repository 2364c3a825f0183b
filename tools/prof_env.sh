# per-kernel durations of the TIMED replays of the headline forward under rocprofv3 for several environment settings on one box
#   bash tools/prof_env.sh "A=1 B=0" "A=1 B=1" ...     (BENCH_ARGS for extra bench arguments)
R=$PWD
for envs in "$@"; do
  rm -rf /tmp/pp; ( cd /tmp; export TMPDIR=/tmp; env $envs ANEMOI_BENCH_SENTINEL=1 rocprofv3 --kernel-trace --stats -d /tmp/pp -o pp -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing $BENCH_ARGS > /tmp/pp.log 2>&1 )
  echo "== $envs  $(python -c 'import sys,json; d=json.loads(open("/tmp/pp.log").read().strip().splitlines()[-1]); print("ms_per_step", round(d["ms_per_step"],4))' 2>/dev/null)"
  python $R/tools/rocprof_summary.py $(find /tmp/pp -name "*.db" | head -1) --timed 2>&1 | head -${TOPN:-14} | cut -c1-70,108-170
done
