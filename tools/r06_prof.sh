# usage (GPU box, repo root): bash tools/r06_prof.sh TAG [bench args] -> timed-replay per-kernel summary (rocprofv3 --kernel-trace --stats, sentinel-bracketed)
R=$PWD; TAG=$1; shift
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pp_$TAG
ANEMOI_BENCH_SENTINEL=1 timeout 280 rocprofv3 --kernel-trace --stats -d /tmp/pp_$TAG -o p -- python $R/bench.py "$@" --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > $OUT/bench_under_rocprof.json 2>/dev/null < /dev/null
d=$(find /tmp/pp_$TAG -name "*.db" | head -1)
[ -n "$d" ] && python $R/tools/rocprof_summary.py $d --timed > $OUT/kernel_trace_summary.txt 2>&1
cd $R; cut -c1-70,108-170 $OUT/kernel_trace_summary.txt | head -24
