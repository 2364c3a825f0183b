# PMC traffic (FETCH_SIZE / WRITE_SIZE in separate rocprofv3 passes, eager forward) for the non-headline configurations:
# writes gpurun_out/r02p/pmc_traffic_<config>.json (copied to profiles/r02_pmc_traffic_<config>.json, which bench.py --config reads)
set -x
R=$PWD
OUT=$R/gpurun_out/r02p
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for c in o96-res6 n320 gnn; do
  rm -rf /tmp/pf_$c /tmp/pw_$c
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf_$c -o pf -- python $R/bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-graph > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pw_$c -o pw -- python $R/bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-graph > /dev/null 2>&1
  python $R/tools/pmc_traffic.py $(find /tmp/pf_$c -name "*.db" | head -1) $(find /tmp/pw_$c -name "*.db" | head -1) $OUT/pmc_traffic_$c.json > $OUT/pmc_traffic_$c.log 2>&1
done
cd $R
cat $OUT/pmc_traffic_*.json
