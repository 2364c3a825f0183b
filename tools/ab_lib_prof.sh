# usage: bash tools/ab_lib_prof.sh CONFIG PATTERN [ALT_LIB] -> mean duration of the kernels matching PATTERN in a 20-step bench of CONFIG under rocprofv3,
# for the library in the tree and (if given) another build of it (ANEMOI_HIP_LIB), on the same box
R=$PWD
for l in tree ${3:+alt}; do
  if [ $l = alt ]; then export ANEMOI_HIP_LIB=$3; else unset ANEMOI_HIP_LIB; fi
  rm -rf /tmp/pp_$l
  ( cd /tmp; export TMPDIR=/tmp; ANEMOI_TORCH_EXT=0 timeout 280 rocprofv3 --kernel-trace --stats -d /tmp/pp_$l -o pp -- python $R/bench.py --config $1 --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1 < /dev/null )
  db=$(find /tmp/pp_$l -name "*.db" | head -1)
  echo "== $l ${ANEMOI_HIP_LIB:-}"
  if [ -n "$db" ]; then python $R/tools/rocprof_summary.py $db 2>&1 < /dev/null | grep -E "$2|TOTAL" | cut -c1-70,108-170; else echo "no trace written"; fi
done
