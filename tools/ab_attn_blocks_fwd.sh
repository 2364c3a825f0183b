# forwards (ms, 40 graph replays) against ANEMOI_ATTN_BLOCKS_PER_CU: usage: bash tools/ab_attn_blocks_fwd.sh "0 10 16" "o96-res6 n320"
export ANEMOI_TORCH_EXT=0
for rep in 1 2; do
for c in $2; do
for b in $1; do
  if [ $b = 0 ]; then unset ANEMOI_ATTN_BLOCKS_PER_CU; else export ANEMOI_ATTN_BLOCKS_PER_CU=$b; fi
  python bench.py --config $c --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("forward '$c' blocks_per_cu '$b'", round(d["ms_per_step"],4))'
done; done; done
