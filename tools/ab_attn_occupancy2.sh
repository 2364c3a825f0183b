export ANEMOI_TORCH_EXT=0
R=$PWD
for v in tree pf2w6 pf1; do for b in 0 6 8 12; do
  if [ $v = tree ]; then unset ANEMOI_HIP_LIB; else export ANEMOI_HIP_LIB=$R/anemoi_core_amd/lib/alt_$v.so; fi
  if [ $b = 0 ]; then unset ANEMOI_ATTN_BLOCKS_PER_CU; else export ANEMOI_ATTN_BLOCKS_PER_CU=$b; fi
  echo "== $v blocks_per_cu $b"
  python tools/kernel_time.py "attention" 300 2>/dev/null | grep -i "fused_edge" | cut -c1-150
  python tools/kernel_time.py "attention" 300 --res 6 2>/dev/null | grep -i "fused_edge" | cut -c1-150
done; done
