# same-box timing-only ablations of gt_attn_fused_edge_fwd_kernel (ANEMOI_ATTN_DBG bits, see csrc/gt_attention.hip): what each part
# of the kernel costs, i.e. the upper bound of any rewrite of that part.  Results of the dbg builds are wrong by construction.
# build first:  for d in 1 2 3 4 7 16 32 39; do bash tools/build_alt.sh dbg$d -DANEMOI_ATTN_DBG=$d; done
# usage: bash tools/ab_attn_ablate.sh [reps]
R=$PWD; export ANEMOI_TORCH_EXT=0
for rep in 1 2; do
for v in tree dbg1 dbg2 dbg3 dbg4 dbg7 dbg16 dbg32 dbg39; do
  if [ $v = tree ]; then unset ANEMOI_HIP_LIB; else export ANEMOI_HIP_LIB=$R/anemoi_core_amd/lib/alt_$v.so; [ -f $ANEMOI_HIP_LIB ] || continue; fi
  echo "== $v"
  python tools/kernel_time.py "attention" ${1:-300} 2>/dev/null | grep -i "fused_edge" | cut -c1-150
  python tools/kernel_time.py "attention" ${1:-300} --res 6 2>/dev/null | grep -i "fused_edge" | cut -c1-150
done; done
