# Round-2 profile refresh (run on the GPU box from the repo root): bench lines + rocprofv3 kernel-trace summaries per BASELINE
# configuration, PMC traffic (FETCH_SIZE / WRITE_SIZE in separate passes) and SQ counters for the headline configuration.
# Outputs land in gpurun_out/r02p/ and are copied into profiles/r02_* by hand.
set -x
R=$PWD
OUT=$R/gpurun_out/r02p
mkdir -p $OUT
python bench.py > $OUT/bench_o96.json 2> $OUT/bench_o96.err
for c in o96-res6 n320 gnn; do python bench.py --config $c --steps 10 --warmup 3 > $OUT/bench_$c.json 2> $OUT/bench_$c.err; done
cd /tmp; export TMPDIR=/tmp
for c in o96 o96-res6 n320 gnn; do
  rm -rf /tmp/p_$c
  rocprofv3 --kernel-trace --stats -d /tmp/p_$c -o p -- python $R/bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > $OUT/bench_under_rocprof_$c.json 2>/dev/null
  python $R/tools/rocprof_summary.py $(find /tmp/p_$c -name "*.db" | head -1) > $OUT/kernel_trace_summary_$c.txt 2>&1
done
# PMC passes (own runs, no tracing domains besides the kernel trace), eager forward so that every launch is a dispatch
rm -rf /tmp/pf /tmp/pw /tmp/psq
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf -o pf -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-graph > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pw -o pw -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-graph > /dev/null 2>&1
python $R/tools/pmc_traffic.py $(find /tmp/pf -name "*.db" | head -1) $(find /tmp/pw -name "*.db" | head -1) $OUT/pmc_traffic.json > $OUT/pmc_traffic.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU -d /tmp/psq -o psq -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-graph > /dev/null 2>&1
DB=$(find /tmp/psq -name "*.db" | head -1)
python $R/tools/pmc_summary.py $DB linear_mfma > $OUT/pmc_sq_linear.txt 2>&1
python $R/tools/pmc_summary.py $DB gt_attn_fused_edge > $OUT/pmc_sq_attention.txt 2>&1
python $R/tools/pmc_summary.py $DB layernorm > $OUT/pmc_sq_layernorm.txt 2>&1
cd $R
head -c 300 $OUT/bench_o96.json
