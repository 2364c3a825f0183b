# panel-height rule and warm-up shares: tree (48-row panels always; warm-up shares follow the workgroup count) / tree with balanced rounds for
# multi-round launches (ANEMOI_CHAIN_BALANCED=1) / the previous library (balanced always, fixed 1/32 shares), same box
export ANEMOI_TORCH_EXT=0
run() { python bench.py --config $1 --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], sys.argv[2], round(d["ms_per_step"],4))' "$1" "$2"; }
for c in ${CONFIGS:-o96 o96-res6 n320}; do for rep in 1 2 3; do
  unset ANEMOI_HIP_LIB ANEMOI_CHAIN_BALANCED; run $c tree
  export ANEMOI_CHAIN_BALANCED=1; run $c balanced; unset ANEMOI_CHAIN_BALANCED
  export ANEMOI_HIP_LIB=$PWD/anemoi_core_amd/lib/alt_old.so; run $c old
done; done
