# write-through stores: the chain's output rows (alt_wt.so = -DANEMOI_CHAIN2_NT=5) and the attention's (ANEMOI_ATTN_OUT_WT=1), same box
export ANEMOI_TORCH_EXT=0
run() { python bench.py --config $1 --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], sys.argv[2], round(d["ms_per_step"],4))' "$1" "$2"; }
for c in o96 o96-res6; do for rep in 1 2 3; do
  unset ANEMOI_HIP_LIB; unset ANEMOI_ATTN_OUT_WT; run $c tree
  export ANEMOI_ATTN_OUT_WT=1; run $c attn_wt; unset ANEMOI_ATTN_OUT_WT
  export ANEMOI_HIP_LIB=$PWD/anemoi_core_amd/lib/alt_wt.so; run $c chain_wt
  export ANEMOI_ATTN_OUT_WT=1; run $c both_wt
done; done
