R=$PWD; export ANEMOI_TORCH_EXT=0
run() { python bench.py --config $1 --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("forward '$1' '$2'", round(d["ms_per_step"],4))'; }
for rep in 1 2; do
unset ANEMOI_HIP_LIB ANEMOI_ATTN_BLOCKS_PER_CU; run o96 tree; run o96-res6 tree
export ANEMOI_HIP_LIB=$R/anemoi_core_amd/lib/alt_fs2.so; run o96 fs2; run o96-res6 fs2
export ANEMOI_ATTN_BLOCKS_PER_CU=6; run o96-res6 fs2-bpc6; unset ANEMOI_ATTN_BLOCKS_PER_CU
done
