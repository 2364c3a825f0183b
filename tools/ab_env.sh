# usage: bash tools/ab_env.sh VAR  -> bench with VAR=0 / VAR=1 alternating on the same box
for rep in 1 2 3; do for v in 0 1; do
  env $1=$v python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("'$1=$v'", round(d["ms_per_step"],4))'
done; done
