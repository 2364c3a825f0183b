for r in 3 4; do for f in 0 1; do
  echo "res $r fold $f"
  ANEMOI_LN_FOLD=$f python bench.py --hidden-res $r --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("ms", round(d["ms_per_step"],4), d.get("components"))'
done; done
