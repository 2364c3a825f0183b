# same-box A/B of the library in the tree against another build of it: bash tools/ab_lib.sh anemoi_core_amd/lib/alt_X.so "o96 o96-res6" [reps]
ALT=$PWD/$1; CONFIGS=${2:-o96}; REPS=${3:-3}
export ANEMOI_TORCH_EXT=0
run() { python bench.py --config $1 --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], sys.argv[2], round(d["ms_per_step"],4))' "$1" "$2"; }
for c in $CONFIGS; do for rep in $(seq $REPS); do
  unset ANEMOI_HIP_LIB; run $c tree
  export ANEMOI_HIP_LIB=$ALT; run $c alt
done; done
