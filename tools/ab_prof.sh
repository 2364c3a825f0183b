# usage: bash tools/ab_prof.sh OTHER_TREE -> per-kernel mean durations of this tree and another checkout on the same box
R=$PWD
for t in . $1; do
  cd $R/$t; rm -rf /tmp/pp; ( cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats -d /tmp/pp -o pp -- python $R/$t/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1 )
  echo "== $t"; python $R/tools/rocprof_summary.py $(find /tmp/pp -name "*.db" | head -1) 2>&1 | head -8 | cut -c1-60,108-160
done
