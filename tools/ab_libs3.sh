# same-box comparison of the tree library and several other builds: bash tools/ab_libs3.sh "o96 o96-res6" 2 alt_a.so alt_b.so ...
CONFIGS=$1; REPS=$2; shift 2
export ANEMOI_TORCH_EXT=0
run() { python bench.py --config $1 --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], sys.argv[2], round(d["ms_per_step"],4))' "$1" "$2"; }
for c in $CONFIGS; do for rep in $(seq $REPS); do
  unset ANEMOI_HIP_LIB; run $c tree
  for l in "$@"; do export ANEMOI_HIP_LIB=$PWD/anemoi_core_amd/lib/$l; run $c $l; done
done; done
