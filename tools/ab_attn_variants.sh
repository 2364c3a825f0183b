# same-box A/B of attention kernel builds: usage: bash tools/ab_attn_variants.sh "tree v1 v2 ..." [reps]   (anemoi_core_amd/lib/alt_<v>.so)
R=$PWD; export ANEMOI_TORCH_EXT=0
for rep in 1 2; do
for v in $1; do
  if [ $v = tree ]; then unset ANEMOI_HIP_LIB; else export ANEMOI_HIP_LIB=$R/anemoi_core_amd/lib/alt_$v.so; [ -f $ANEMOI_HIP_LIB ] || continue; fi
  echo "== $v"
  python tools/kernel_time.py "attention" ${2:-300} 2>/dev/null | grep -i "fused_edge" | cut -c1-150
  python tools/kernel_time.py "attention" ${2:-300} --res 6 2>/dev/null | grep -i "fused_edge" | cut -c1-150
done; done
