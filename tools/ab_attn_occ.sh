# same-box A/B of the attention kernel's register budget: tree library (96 VGPR, 5 waves/SIMD) vs alt_mw6.so (80 VGPR, 6 waves/SIMD),
# kernel train time at 10 242 / 40 962 destinations and whole forwards.  usage: bash tools/ab_attn_occ.sh
R=$PWD; export ANEMOI_TORCH_EXT=0
for rep in 1 2; do
for v in tree mw6:0 mw6:6 mw6:5; do
  lib=${v%%:*}; bpc=${v##*:}
  if [ $lib = tree ]; then unset ANEMOI_HIP_LIB; else export ANEMOI_HIP_LIB=$R/anemoi_core_amd/lib/alt_$lib.so; fi
  if [ "$bpc" = "$v" ] || [ "$bpc" = 0 ]; then unset ANEMOI_ATTN_BLOCKS_PER_CU; else export ANEMOI_ATTN_BLOCKS_PER_CU=$bpc; fi
  echo "== $v"
  python tools/kernel_time.py "attention" 300 2>/dev/null | grep -i "attn\|attention" | cut -c1-150
  python tools/kernel_time.py "attention" 300 --res 6 2>/dev/null | grep -i "processor" | cut -c1-150
  python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("o96 forward", round(d["ms_per_step"],4))'
done; done
