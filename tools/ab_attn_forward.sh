# same-box A/B of whole forwards on alternative library builds: usage: bash tools/ab_attn_forward.sh "tree v1 ..." "o96 o96-res6 n320"
R=$PWD; export ANEMOI_TORCH_EXT=0
for rep in ${REPS:-1 2}; do
for c in $2; do
for v in $1; do
  if [ $v = tree ]; then unset ANEMOI_HIP_LIB; else export ANEMOI_HIP_LIB=$R/anemoi_core_amd/lib/alt_$v.so; [ -f $ANEMOI_HIP_LIB ] || continue; fi
  python bench.py --config $c --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("forward '$c' '$v'", round(d["ms_per_step"],4))'
done; done; done
