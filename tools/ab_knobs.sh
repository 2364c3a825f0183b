# same-box A/B of environment knobs in the headline forward: bash tools/ab_knobs.sh "A=1" "B=1 C=2" ...   (the empty setting "X=0" = defaults)
run() { env $1 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-timing $BENCH_ARGS 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],4))' "$1"; }
for rep in 1 2 3; do for e in "$@"; do run "$e"; done; done
