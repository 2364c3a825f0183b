# usage (GPU box, repo root): bash tools/r06_pmc_sq.sh TAG [CONFIG] [NAME] [extra bench.py arguments] -> SQ counter passes of the eager forward (default o96;
# NAME names the output files when extra arguments change the workload, e.g. `r06 o96 hres4 --hidden-res 4`), per-kernel averages for the
# role-split chain, the fused attention and the mapper-side GEMMs (tools/pmc_summary.py): pass A = busy / wait / MFMA-busy cycles, pass B = instruction
# mix.  Counters in their own passes (--kernel-trace only beside --pmc).
R=$PWD
OUT=$R/gpurun_out/$1
c=${2:-o96}
name=${3:-$c}
shift; shift; shift
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u > $OUT/sq_counters_available.txt
PA="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU"
PB="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_BUSY_CYCLES"
PC="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES"
i=0
for P in "$PA" "$PB" "$PC"; do
  i=$((i+1))
  rm -rf /tmp/psq_$i; rm -f $OUT/pmc_sq_${name}_pass$i.txt
  timeout 280 rocprofv3 --kernel-trace --pmc $P -d /tmp/psq_$i -o psq -- python $R/bench.py --config $c "$@" --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-graph > $OUT/pmc_sq_pass$i.log 2>&1 < /dev/null
  DB=$(find /tmp/psq_$i -name "*.db" | head -1)
  if [ -n "$DB" ]; then
    for k in gt_chain2 gt_attn_fused gt_rowchain gt_cluster linear_mfma gnn_edge_chain gnn_node_chain segment_sum; do python $R/tools/pmc_summary.py $DB $k >> $OUT/pmc_sq_${name}_pass$i.txt 2>&1; done
  else
    tail -5 $OUT/pmc_sq_pass$i.log
  fi
done
cd $R
cat $OUT/pmc_sq_${name}_pass1.txt | cut -c1-160
