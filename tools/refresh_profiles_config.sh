# usage (GPU box, repo root): [ROUND=r05] bash tools/refresh_profiles_config.sh CONFIG TAG -> PMC traffic passes, bench line and timed-replay kernel summary of ONE
# configuration (the per-configuration part of tools/refresh_profiles_r03.sh), written to gpurun_out/TAG/ and, for the PMC files, to profiles/
# on the box so that the bench line taken afterwards carries `traffic`.
R=$PWD
ROUND=${ROUND:-r05}
c=$1
OUT=$R/gpurun_out/$2
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pf_$c /tmp/pw_$c /tmp/p_$c
timeout 280 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf_$c -o pf -- python $R/bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-graph > /dev/null 2>&1 < /dev/null
timeout 280 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pw_$c -o pw -- python $R/bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-graph > /dev/null 2>&1 < /dev/null
s=$([ $c = o96 ] && echo "" || echo "_$c")
f=$(find /tmp/pf_$c -name "*.db" | head -1); w=$(find /tmp/pw_$c -name "*.db" | head -1)
if [ -n "$f" ] && [ -n "$w" ]; then
  python $R/tools/pmc_traffic.py $f $w $OUT/pmc_traffic$s.json > $OUT/pmc_traffic$s.log 2>&1 < /dev/null
  cp $OUT/pmc_traffic$s.json $R/profiles/${ROUND}_pmc_traffic$s.json
  cp $OUT/pmc_traffic${s}_detail.json $R/profiles/${ROUND}_pmc_traffic${s}_detail.json
fi
ANEMOI_BENCH_SENTINEL=1 timeout 280 rocprofv3 --kernel-trace --stats -d /tmp/p_$c -o p -- python $R/bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > $OUT/bench_under_rocprof_$c.json 2>/dev/null < /dev/null
d=$(find /tmp/p_$c -name "*.db" | head -1)
if [ -n "$d" ]; then python $R/tools/rocprof_summary.py $d --timed > $OUT/kernel_trace_summary_$c.txt 2>&1 < /dev/null; cp $OUT/kernel_trace_summary_$c.txt $R/profiles/${ROUND}_kernel_trace_summary_$c.txt; fi
cd $R
timeout 600 python bench.py --config $c --steps 10 --warmup 3 > $OUT/bench_$c.json 2> $OUT/bench_$c.err < /dev/null
head -c 1500 $OUT/bench_$c.json; echo; head -12 $OUT/kernel_trace_summary_$c.txt | cut -c1-70,108-170
