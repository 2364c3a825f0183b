# Round-3 profile refresh (GPU box, repo root).  Order matters: the PMC traffic passes come FIRST and are copied into profiles/
# on the box, so that the bench lines taken afterwards carry `traffic` / `traffic_over_algorithmic` (hygiene b).
# The kernel-trace summaries keep only the dispatches of the TIMED replays (ANEMOI_BENCH_SENTINEL=1 + rocprof_summary --timed).
# Outputs: gpurun_out/r03p/ (merged back), to be copied into profiles/r03_*.
set -x
R=$PWD
OUT=$R/gpurun_out/r03p
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for c in o96 o96-res6 n320 gnn; do
  rm -rf /tmp/pf_$c /tmp/pw_$c
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf_$c -o pf -- python $R/bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-graph > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pw_$c -o pw -- python $R/bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-graph > /dev/null 2>&1
  s=$([ $c = o96 ] && echo "" || echo "_$c")
  python $R/tools/pmc_traffic.py $(find /tmp/pf_$c -name "*.db" | head -1) $(find /tmp/pw_$c -name "*.db" | head -1) $OUT/pmc_traffic$s.json > $OUT/pmc_traffic$s.log 2>&1
  cp $OUT/pmc_traffic$s.json $R/profiles/r03_pmc_traffic$s.json
  cp $OUT/pmc_traffic${s}_detail.json $R/profiles/r03_pmc_traffic${s}_detail.json
done
cd $R
python bench.py > $OUT/bench_o96.json 2> $OUT/bench_o96.err
for c in o96-res6 n320 gnn; do python bench.py --config $c --steps 10 --warmup 3 > $OUT/bench_$c.json 2> $OUT/bench_$c.err; done
cd /tmp
for c in o96 o96-res6 n320 gnn; do
  rm -rf /tmp/p_$c
  ANEMOI_BENCH_SENTINEL=1 rocprofv3 --kernel-trace --stats -d /tmp/p_$c -o p -- python $R/bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > $OUT/bench_under_rocprof_$c.json 2>/dev/null
  python $R/tools/rocprof_summary.py $(find /tmp/p_$c -name "*.db" | head -1) --timed > $OUT/kernel_trace_summary_$c.txt 2>&1
done
rm -rf /tmp/psq
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU -d /tmp/psq -o psq -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-graph > /dev/null 2>&1
DB=$(find /tmp/psq -name "*.db" | head -1)
python $R/tools/pmc_summary.py $DB linear_mfma > $OUT/pmc_sq_linear.txt 2>&1
python $R/tools/pmc_summary.py $DB gt_attn_fused_edge > $OUT/pmc_sq_attention.txt 2>&1
cd $R
# model-parallel evidence on the one GPU: per-rank floors (peers leave before the timing), the exchange kernel's own cost
for w in 2 4 8; do timeout 600 python tools/rank_floor.py --world $w --hidden-res 5 --wire ipc 2> $OUT/floor_w${w}_r5.err | tail -1 > $OUT/floor_w${w}_r5.json; done
timeout 600 python tools/rank_floor.py --world 8 --hidden-res 6 --wire ipc 2> $OUT/floor_w8_r6.err | tail -1 > $OUT/floor_w8_r6.json
timeout 300 python tools/peer_latency.py --loopback 2>&1 | grep peer_latency > $OUT/peer_latency.txt
timeout 300 python tools/peer_latency.py --loopback --rows 512 2>&1 | grep peer_latency >> $OUT/peer_latency.txt
timeout 300 python tools/peer_latency.py --world 2 2>&1 | grep peer_latency >> $OUT/peer_latency.txt
ANEMOI_BENCH_TRANSPORT=ipc timeout 600 python bench.py --gpus 8 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_8ranks_one_gpu_ipc.json 2> $OUT/bench_8ranks_one_gpu_ipc.err
head -c 400 $OUT/bench_o96.json; cat $OUT/floor_w*_r*.json $OUT/peer_latency.txt
