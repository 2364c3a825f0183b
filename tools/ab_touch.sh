# the L2 warm-up's coverage: tree (every workgroup of an XCD takes 1 / n of a segment's lines, n = the launch's workgroups on that XCD) against
# alt_old.so (a fixed 1/32 share), at the default panel height (10 242 rows = 250 panels of 41) and at 48 rows (214 panels), same box
export ANEMOI_TORCH_EXT=0
run() { python bench.py --config $1 --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], sys.argv[2], round(d["ms_per_step"],4))' "$1" "$2"; }
for rep in 1 2 3; do
  unset ANEMOI_HIP_LIB ANEMOI_CHAIN_ROWS; run o96 tree
  export ANEMOI_HIP_LIB=$PWD/anemoi_core_amd/lib/alt_old.so; run o96 old
  export ANEMOI_CHAIN_ROWS=48; run o96 old_rows48
  unset ANEMOI_HIP_LIB; run o96 tree_rows48
done
