for rep in 1 2; do
for e in "X=0" "ANEMOI_CHAIN2_DBG=8" "ANEMOI_CHAIN2_DBG=8 ANEMOI_CHAIN2_WARM=0" "ANEMOI_CHAIN2_WARM=0"; do
  echo "== $e"
  env $e python tools/kernel_time.py "chain2" 200 --res 6 2>/dev/null | grep chain2 | cut -c1-160
  env $e python tools/kernel_time.py "chain2" 200 2>/dev/null | grep chain2 | cut -c1-160
done; done
