# usage: bash tools/ab_env2.sh VAR "CONFIGS" REPS -> bench with VAR=0 / VAR=1 alternating on the same box, per configuration
V=$1; CONFIGS=${2:-o96}; REPS=${3:-3}
for c in $CONFIGS; do for rep in $(seq $REPS); do for v in 0 1; do
  env $V=$v python bench.py --config $c --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], sys.argv[2], round(d["ms_per_step"],4))' $c $V=$v
done; done; done
