# usage (GPU box, repo root): bash tools/small_mesh_prof.sh [RES ...] -> forward time and per-kernel table (timed replays) of the O96 model with a small hidden
# mesh (res 3 = 642 nodes, res 4 = 2 562): the regime of one rank's share of a sharded mesh.  Environment switches pass through.
R=$PWD
for r in ${@:-3 4}; do
rm -rf /tmp/ps$r
( cd /tmp; export TMPDIR=/tmp; ANEMOI_BENCH_SENTINEL=1 timeout 280 rocprofv3 --kernel-trace --stats -d /tmp/ps$r -o p -- python $R/bench.py --hidden-res $r --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > /tmp/ps$r.json 2>/dev/null < /dev/null )
d=$(find /tmp/ps$r -name "*.db" | head -1)
echo "== hidden res $r: $(python -c "import json;print(json.loads(open('/tmp/ps$r.json').read().strip().splitlines()[-1])['ms_per_step'])") ms"
if [ -n "$d" ]; then python $R/tools/rocprof_summary.py $d --timed 2>&1 < /dev/null | cut -c1-75,108-170 | head -${SMALL_MESH_ROWS:-22}; fi
done
