# kernel sequence (start, duration) of the LAST timed forward under rocprofv3 for an environment setting:  bash tools/seq_env.sh "A=1 B=1" <kernels per forward>
R=$PWD
rm -rf /tmp/pp; ( cd /tmp; export TMPDIR=/tmp; env $1 rocprofv3 --kernel-trace --stats -d /tmp/pp -o pp -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing $BENCH_ARGS > /tmp/pp.log 2>&1 )
cd $R/tools && python rocprof_sequence.py $(find /tmp/pp -name "*.db" | head -1) $2
