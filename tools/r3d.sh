# Round-3 measurement pass D (GPU box, repo root): new tests, small-mesh LayerNorm fold, rank floors, peer latency, attention order A/B
set -x
R=$PWD
OUT=$R/gpurun_out/r3d
mkdir -p $OUT
export ANEMOI_PEER_TIMEOUT_S=15
(timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_distributed_gpu.py -m gpu -x -q 2>&1 | tail -15) > $OUT/tests_a.log 2>&1
(timeout 900 python -m pytest tests/test_config3_sharded_gpu.py tests/test_training_gpu.py -m gpu -x -q -k "bench_entry or edge_pre" 2>&1 | grep -v "amdgpu.ids\|Gloo" | tail -30) > $OUT/tests_bench8.log 2>&1
# LayerNorm fold on small meshes (res 3 = 642, res 4 = 2562 hidden nodes): old gate (4096 rows) against no gate
for r in 3 4; do for m in 4096 0; do
  echo "res $r fold_min_rows $m" >> $OUT/small_mesh.log
  ANEMOI_LN_FOLD_MIN_ROWS=$m python bench.py --hidden-res $r --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("ms", round(d["ms_per_step"],4))' >> $OUT/small_mesh.log
done; done
# rank floors (one rank's share, no-op wire and the ipc wire's kernels without the wait)
for m in 4096 0; do
  ANEMOI_LN_FOLD_MIN_ROWS=$m timeout 600 python tools/rank_floor.py --world 8 --hidden-res 5 --wire ipc > $OUT/floor_w8_r5_fold$m.json 2> $OUT/floor_w8_r5_fold$m.err
done
ANEMOI_LN_FOLD_MIN_ROWS=0 timeout 600 python tools/rank_floor.py --world 8 --hidden-res 6 --wire ipc > $OUT/floor_w8_r6_fold0.json 2> $OUT/floor_w8_r6.err
timeout 300 python tools/peer_latency.py --loopback > $OUT/peer_latency.log 2>&1
timeout 300 python tools/peer_latency.py --loopback --rows 512 >> $OUT/peer_latency.log 2>&1
timeout 300 python tools/peer_latency.py --world 2 >> $OUT/peer_latency.log 2>&1
timeout 300 python tools/peer_latency.py --world 3 >> $OUT/peer_latency.log 2>&1
# attention work order / write-through output stores at res 6 (the processor kernel alone, then the forward)
for o in 0 1; do for w in 0 1; do
  echo "order $o out_wt $w" >> $OUT/attn_order.log
  ANEMOI_ATTN_ORDER=$o ANEMOI_ATTN_OUT_WT=$w python tools/kernel_time.py gt_attention 200 --res 6 >> $OUT/attn_order.log 2>&1
done; done
for o in 0 1; do
  echo "forward o96-res6 order $o" >> $OUT/attn_order.log
  ANEMOI_ATTN_ORDER=$o python bench.py --config o96-res6 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("ms", round(d["ms_per_step"],4))' >> $OUT/attn_order.log
done
cd /tmp; export TMPDIR=/tmp
for o in 0 1; do
  rm -rf /tmp/pf_o$o
  ANEMOI_RUN_KERNEL_RES=6 ANEMOI_ATTN_ORDER=$o rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf_o$o -o pf -- python $R/tools/run_kernel.py gt_attention 10 > /dev/null 2>&1
  python - <<EOF >> $OUT/attn_order.log 2>&1
import sqlite3, glob
db = glob.glob("/tmp/pf_o$o/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
ix = {n: i for i, n in enumerate(cols)}
nm = "kernel_name" if "kernel_name" in ix else [n for n in cols if "kernel" in n and "name" in n][0]
v = [r[ix["value"]] for r in c.execute("select * from counters_collection") if r[ix["counter_name"]] == "FETCH_SIZE" and "gt_attn_fused_edge" in r[ix[nm]]]
print("order $o: FETCH_SIZE x2 corrected per launch = %.1f MB over %d launches" % (sum(v) / len(v) * 1024 * 2 / 1e6, len(v)))
EOF
done
cd $R
tail -5 $OUT/tests_a.log; tail -5 $OUT/tests_bench8.log; cat $OUT/small_mesh.log $OUT/peer_latency.log $OUT/attn_order.log; cat $OUT/floor_w8_r5_fold*.json
