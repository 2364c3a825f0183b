# same-box A/B of the layer chain in the headline forward: off (launch per GEMM) / chain without / with the L2 warm-up (the warm-up switch needs
# the experiments build: ANEMOI_HIP_LIB=anemoi_core_amd/lib/libanemoi_hip_exp.so)
#   bash tools/ab_chain.sh        (BENCH_ARGS for extra bench arguments)
run() { env "$@" python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-timing $BENCH_ARGS 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],4))' "$*"; }
for rep in 1 2 3; do
  run ANEMOI_LAYER_CHAIN=0
  run ANEMOI_LAYER_CHAIN=1 ANEMOI_CHAIN2_WARM=0
  run ANEMOI_LAYER_CHAIN=1 ANEMOI_CHAIN2_WARM=1
done
