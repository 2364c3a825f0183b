export ANEMOI_TORCH_EXT=0
for rep in 1 2; do
for b in 0 4 5 6 7 8 10; do
  if [ $b = 0 ]; then unset ANEMOI_ATTN_BLOCKS_PER_CU; else export ANEMOI_ATTN_BLOCKS_PER_CU=$b; fi
  echo "== blocks_per_cu $b"
  python tools/kernel_time.py "attention" 300 2>/dev/null | grep -i "fused_edge" | cut -c1-150
  python tools/kernel_time.py "attention" 300 --res 6 2>/dev/null | grep -i "fused_edge" | cut -c1-150
done; done
