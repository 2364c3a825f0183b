for rep in 1 2 3; do for v in old nt; do
  ANEMOI_HIP_LIB=$PWD/anemoi_core_amd/lib/alt_$v.so python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("'$v'", round(d["ms_per_step"],4))'
done; done
