# usage (GPU box, repo root): bash tools/pmc_sq_chain.sh TAG -> SQ counters (waves, busy / wait cycles, instructions, MFMA busy cycles) of the row-resident
# chain kernels, per kernel averages (tools/pmc_summary.py), from eager forwards of the GNN configuration and of the O96 GraphTransformer with
# ANEMOI_LAYER_CHAIN=1.  Counters in their own pass (--kernel-trace only beside --pmc).
R=$PWD
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
PMC="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU"
rm -rf /tmp/psq_gnn /tmp/psq_gt
timeout 280 rocprofv3 --kernel-trace --pmc $PMC -d /tmp/psq_gnn -o psq -- python $R/bench.py --config gnn --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-graph > /dev/null 2>&1 < /dev/null
DB=$(find /tmp/psq_gnn -name "*.db" | head -1)
[ -n "$DB" ] && python $R/tools/pmc_summary.py $DB gnn_ > $OUT/pmc_sq_gnn_chain.txt 2>&1
ANEMOI_LAYER_CHAIN=1 timeout 280 rocprofv3 --kernel-trace --pmc $PMC -d /tmp/psq_gt -o psq -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-graph > /dev/null 2>&1 < /dev/null
DB=$(find /tmp/psq_gt -name "*.db" | head -1)
[ -n "$DB" ] && python $R/tools/pmc_summary.py $DB gt_chain > $OUT/pmc_sq_gt_chain.txt 2>&1
cd $R
head -40 $OUT/pmc_sq_gnn_chain.txt | cut -c1-160; head -12 $OUT/pmc_sq_gt_chain.txt | cut -c1-160
